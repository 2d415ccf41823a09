#!/usr/bin/env python3
"""Regenerates tests/golden/reference_known_answers.json.

The reference (Julia) cannot run in the build image, and its tests hold no binary fixtures: every expected
value is a closed-form result on `1:N` / `string.(1:N)` data checked against DataFrames.jl
(SURVEY.md §4, §8c).  This script re-derives those expected values with plain Python/numpy indexing —
the stand-in for DataFrames.jl — citing the reference test each case comes from.  Inputs are described
by a tiny spec ("iota", "iota_str", "arange_f") so the JSON stays small; stages use the same encoding as
tests/helpers.apply_stages with predicates named by key into PREDICATES below (tests/golden_cases.py
builds the IR for each key).

    python tests/golden/make_golden.py   # rewrites the JSON next to this file
"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def iota(n):
    return np.arange(1, n + 1, dtype=np.int64)


cases = []


def case(name, ref, table, block_sizes, stages, proj, expect_rows=None, expect_cols=None, note=""):
    cases.append(dict(name=name, ref=ref, table=table, block_sizes=block_sizes, stages=stages, proj=proj,
                      expect_rows=None if expect_rows is None else [int(x) for x in expect_rows],
                      expect_cols=expect_cols, note=note))


# ---- test/selection.jl -------------------------------------------------------------------------------
a100 = iota(100)
T100 = {"a": ["iota", 100], "b": ["iota_times", 100, 5]}
case("sel_range_of_range", "test/selection.jl:40-49", T100, [50, 100, 7], [["range", 5, 1, 20], ["range", 3, 1, 4]], None, expect_rows=[7, 8])
rows = [r for r in range(10, 61) if 65 > r > 34][14:18]   # survivors of (10:60, 65>a>34) numbered 1.., keep 15:18
case("sel_range_pred_range", "test/selection.jl:51-72", T100, [100, 50, 7],
     [["range", 10, 1, 60], ["pred", "65>a>34"], ["range", 15, 1, 18]], None, expect_rows=rows, note="== 49:52")
assert rows == [49, 50, 51, 52]
mask = (65 > a100) & (a100 > 34) & ((a100 * 5) % 10 == 0)
case("sel_fused_predicates", "test/selection.jl:74-106", T100, [50, 100], [["pred", "65>a>34"], ["pred", "b%10==0"]], None,
     expect_rows=np.nonzero(mask)[0] + 1, expect_cols={"a": a100[mask].tolist(), "b": (a100 * 5)[mask].tolist()})

# ---- test/broadcast.jl, test/projection.jl -----------------------------------------------------------
TB = {"a": ["iota", 100], "b": ["iota_str", 100], "c": ["arange_f", 0.5, 0.5, 100]}
c100 = 0.5 * iota(100)
sel = np.arange(0, 100, 10)
case("bc_nested_on_mask", "test/broadcast.jl:29-52", TB, [100, 30], [["range", 1, 10, 100]], [["r", "a+(a+c)"]], expect_rows=sel + 1,
     expect_cols={"r": (a100[sel] * 2 + c100[sel]).tolist()}, note="test_c = test_func2(a, test_func2(a, c)); the test expects a .* 2 .+ c")
case("bc_scalar_arg", "test/broadcast.jl:54-61", TB, [100], [["range", 1, 10, 100]], [["r", "a+20"]], expect_rows=sel + 1,
     expect_cols={"r": (a100[sel] + 20).tolist()})
case("bc_in_set", "test/broadcast.jl:63-71", TB, [100], [["range", 1, 10, 100]], [["r", "in(a,[1,11,21])"]], expect_rows=sel + 1,
     expect_cols={"r": [bool(v in (1, 11, 21)) for v in a100[sel]]})
case("proj_gather_and_computed", "test/projection.jl:57-80", TB, [100], [["range", 1, 10, 100]], [["a", "a"], ["b", "a*2"]], expect_rows=sel + 1,
     expect_cols={"a": a100[sel].tolist(), "b": (a100[sel] * 2).tolist()})

# ---- test/view.jl (a = c = 1:1000, b = string.(1:1000), block_size = 100) ------------------------------
N = 1000
a = iota(N)
b = [str(i) for i in range(1, N + 1)]
TV = {"a": ["iota", N], "b": ["iota_str", N], "c": ["iota", N]}
case("view_identity", "test/view.jl:30-39", TV, [100, 65536], [["range", 1, 1, 1000]], None, expect_rows=a,
     expect_cols={"a": a.tolist(), "b": b, "c": a.tolist()})
m = a % 50 == 0
case("view_mod50", "test/view.jl:41-43", TV, [100], [["range", 1, 1, 1000], ["pred", "a%50==0"]], None, expect_rows=a[m],
     expect_cols={"a": a[m].tolist(), "b": [b[i] for i in np.nonzero(m)[0]], "c": a[m].tolist()}, note="20 rows")
m2 = m & (a < 930)
case("view_mod50_lt930_div", "test/view.jl:46-50", TV, [100], [["range", 1, 1, 1000], ["pred", "a%50==0"], ["pred", "c<930"]], [["a", "a/50"]],
     expect_rows=a[m2], expect_cols={"a": (a[m2] / 50).tolist()}, note="1.0 ... 18.0")
case("view_proj_pairs", "test/view.jl:60-62", TV, [100], [], [["a", "a"], ["c", "c*2"]], expect_rows=a, expect_cols={"a": a.tolist(), "c": (a * 2).tolist()})
case("view_index_vector", "test/view.jl:80-82", TV, [100], [["idx", [1, 200]]], [["c", "c"]], expect_rows=[1, 200], expect_cols={"c": [1, 200]})
case("view_single_row", "test/view.jl:76-78", TV, [100], [["int", 1]], [["c", "c"]], expect_rows=[1], expect_cols={"c": [1]})
case("view_end_minus_10", "test/view.jl:113-116", TV, [100], [["range", 990, 1, 1000]], [["e", "a"]], expect_rows=np.arange(990, 1001),
     expect_cols={"e": list(range(990, 1001))}, note="end-10:end with end = 1000")
case("view_scalar_10", "test/view.jl:127", TV, [100], [["int", 10]], [["a", "a"]], expect_rows=[10], expect_cols={"a": [10]})

# ---- test/range_indexing.jl (intended range semantics; file is not in runtests) -------------------------
case("range_5_60", "test/range_indexing.jl:19", TV, [100], [["range", 5, 1, 60]], None, expect_rows=np.arange(5, 61))
case("range_5_300", "test/range_indexing.jl:20", TV, [100], [["range", 5, 1, 300]], None, expect_rows=np.arange(5, 301))
case("range_5_300_1000", "test/range_indexing.jl:21", TV, [100], [["range", 5, 300, 1000]], None, expect_rows=np.arange(5, 1001, 300))
case("range_vector_table_order", "test/range_indexing.jl:22", TV, [100], [["idx", [1, 200, 20]]], None, expect_rows=[1, 20, 200], note="quirk Q3: table order wins")
case("range_end_minus_20", "test/range_indexing.jl:27", TV, [100], [["range", 980, 1, 1000]], None, expect_rows=np.arange(980, 1001))

# ---- test/columnbroadcast.jl ---------------------------------------------------------------------------
case("cb_a_plus_20_first20", "test/columnbroadcast.jl:28-29", TV, [100], [["range", 1, 1, 20]], [["r", "a+20"]], expect_rows=np.arange(1, 21),
     expect_cols={"r": (a[:20] + 20).tolist()})
case("cb_a_times_a_minus_20", "test/columnbroadcast.jl:31", TV, [100], [["range", 1, 1, 20]], [["r", "a*a-20"]], expect_rows=np.arange(1, 21),
     expect_cols={"r": (a[:20] * a[:20] - 20).tolist()})
case("cb_a_times_c", "test/columnbroadcast.jl:32", TV, [100], [], [["r", "a*c"]], expect_rows=a, expect_cols={"r": (a * a).tolist()})
case("cb_a_eq_10", "test/columnbroadcast.jl:33", TV, [100], [], [["r", "a==10"]], expect_rows=a, expect_cols={"r": (a == 10).tolist()})
m3 = (300 >= a) & (a >= 10)
case("cb_chain_300_ge_a_ge_10", "test/columnbroadcast.jl:45-48", TV, [100], [["pred", "300>=a>=10"]], None, expect_rows=a[m3], note="291 rows")
assert m3.sum() == 291
m4 = m3 & np.array([s.startswith("1") for s in b])
case("cb_then_startswith_1", "test/columnbroadcast.jl:50-53", TV, [100], [["pred", "300>=a>=10"], ["pred", "startswith(b,'1')"]], None, expect_rows=a[m4],
     expect_cols={"a": a[m4].tolist(), "b": [b[i] for i in np.nonzero(m4)[0]], "c": a[m4].tolist()}, note="110 rows")
assert m4.sum() == 110
case("cb_view_of_columns", "test/columnbroadcast.jl:55-60", TV, [100], [], [["a", "a*3"], ["g", "a*c"]], expect_rows=a,
     expect_cols={"a": (a * 3).tolist(), "g": (a * a).tolist()})

# ---- test/column.jl --------------------------------------------------------------------------------------
case("col_90_110", "test/column.jl:33-37", TV, [100], [["range", 90, 1, 110]], [["a", "a"]], expect_rows=np.arange(90, 111), expect_cols={"a": list(range(90, 111))},
     note="col2[1] == 90, col2[12] == 101")
case("col_a_plus_c2", "test/column.jl:39-40", TV, [100], [], [["a", "a+c*2"]], expect_rows=a, expect_cols={"a": (a + a * 2).tolist()})
case("col_a_times_4", "test/column.jl:42-43", TV, [100], [], [["a", "a*4"]], expect_rows=a, expect_cols={"a": (a * 4).tolist()})

# ---- test/flat_strings.jl ----------------------------------------------------------------------------------
FS = ["1", "222", "32", "44", "335", "11116", "312313127", "444", "assadf", "bvxvbx"]
TF = {"s": ["strings", FS]}
case("fsv_range_3_5", "test/flat_strings.jl:67-69", TF, [4, 100], [["range", 3, 1, 5]], None, expect_rows=[3, 4, 5], expect_cols={"s": FS[2:5]})
case("fsv_step_1_2_10", "test/flat_strings.jl:71-73", TF, [4, 100], [["range", 1, 2, 10]], None, expect_rows=[1, 3, 5, 7, 9], expect_cols={"s": FS[0:10:2]})
ms = [s.startswith("3") for s in FS]
case("fsv_startswith_3", "test/flat_strings.jl:77-78", TF, [4, 100], [["pred", "startswith(s,'3')"]], None, expect_rows=[i + 1 for i, v in enumerate(ms) if v],
     expect_cols={"s": [s for s, v in zip(FS, ms) if v]})
FM = ["1", "222", None, "44", "335", "11116", None, "444", "assadf", "bvxvbx"]
TM = {"s": ["strings", FM]}
case("fsv_missing_range", "test/flat_strings.jl:80-84", TM, [4, 100], [["range", 3, 1, 5]], None, expect_rows=[3, 4, 5], expect_cols={"s": FM[2:5]})
case("fsv_missing_step", "test/flat_strings.jl:86-92", TM, [4, 100], [["range", 1, 2, 10]], None, expect_rows=[1, 3, 5, 7, 9], expect_cols={"s": FM[0:10:2]})
case("fsv_missing_empty", "test/flat_strings.jl:88-89", TM, [4, 100], [["idx", []]], None, expect_rows=[], expect_cols={"s": []})
case("fsv_ismissing", "test/flat_strings.jl:93-95", TM, [4, 100], [["pred", "ismissing(s)"]], None, expect_rows=[3, 7], expect_cols={"s": [None, None]})
case("fsv_not_missing", "test/flat_strings.jl:95", TM, [4, 100], [["pred", "!ismissing(s)"]], None, expect_rows=[1, 2, 4, 5, 6, 8, 9, 10],
     expect_cols={"s": [s for s in FM if s is not None]})

# ---- test/missings.jl ------------------------------------------------------------------------------------------
MI = [1, None, 2, 3, None, 5, 6, None, 10, 11, None]
case("missing_int_roundtrip", "test/missings.jl:4-10", {"m": ["nullable_i64", MI]}, [4, 64, 100], [], None, expect_rows=list(range(1, 12)), expect_cols={"m": MI})

out = os.path.join(HERE, "reference_known_answers.json")
with open(out, "w") as f:
    json.dump(dict(comment="hand/numpy-derived known answers of the reference's own tests; see make_golden.py", cases=cases), f, indent=0)
print(f"wrote {len(cases)} cases to {out}")
