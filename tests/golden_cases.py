"""Loader for tests/golden/reference_known_answers.json: table specs -> columns, predicate keys -> IR."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load_cases():
    with open(os.path.join(HERE, "golden", "reference_known_answers.json")) as f:
        return json.load(f)["cases"]


def build_columns(spec: dict) -> dict:
    cols = {}
    for name, s in spec.items():
        kind = s[0]
        if kind == "iota":
            cols[name] = np.arange(1, s[1] + 1, dtype=np.int64)
        elif kind == "iota_times":
            cols[name] = np.arange(1, s[1] + 1, dtype=np.int64) * s[2]
        elif kind == "iota_str":
            cols[name] = [str(i) for i in range(1, s[1] + 1)]
        elif kind == "arange_f":
            cols[name] = s[1] + s[2] * np.arange(0, s[3], dtype=np.float64)
        elif kind == "strings":
            cols[name] = list(s[1])
        elif kind == "nullable_i64":
            vals = np.array([0 if v is None else v for v in s[1]], np.int64)
            cols[name] = np.ma.masked_array(vals, mask=[v is None for v in s[1]])
        else:
            raise KeyError(kind)
    return cols


def expr_for(key: str, names: list):
    """The IR each predicate / projection key of the golden file stands for (Julia source in the key)."""
    from dfdb import ir
    c = {n: ir.col(i) for i, n in enumerate(names)}
    a, b, cc, s = c.get("a"), c.get("b"), c.get("c"), c.get("s")
    table = {
        "65>a>34": lambda: (65 > a) & (a > 34),
        "b%10==0": lambda: b % 10 == 0,
        "a+(a+c)": lambda: a + (a + cc),
        "a+20": lambda: a + 20,
        "in(a,[1,11,21])": lambda: ir.isin(a, [1, 11, 21]),
        "a": lambda: a, "c": lambda: cc,
        "a*2": lambda: a * 2,
        "a%50==0": lambda: a % 50 == 0,
        "c<930": lambda: cc < 930,
        "a/50": lambda: a / 50,
        "c*2": lambda: cc * 2,
        "a*a-20": lambda: a * a - 20,
        "a*c": lambda: a * cc,
        "a==10": lambda: a == 10,
        "300>=a>=10": lambda: (300 >= a) & (a >= 10),
        "startswith(b,'1')": lambda: ir.startswith(b, "1"),
        "a*3": lambda: a * 3,
        "a+c*2": lambda: a + cc * 2,
        "a*4": lambda: a * 4,
        "startswith(s,'3')": lambda: ir.startswith(s, "3"),
        "ismissing(s)": lambda: ir.ismissing(s),
        "!ismissing(s)": lambda: ~ir.ismissing(s),
    }
    return table[key]()


def stages_for(case: dict, names: list):
    out = []
    for st in case["stages"]:
        if st[0] == "pred":
            out.append(("pred", expr_for(st[1], names)))
        elif st[0] == "idx":
            out.append(("idx", list(st[1])))
        else:
            out.append(tuple(st))
    return out


def proj_for(case: dict, names: list):
    if case["proj"] is None:
        return None
    return [(n, expr_for(k, names)) for n, k in case["proj"]]


def check_columns(case: dict, names: list, got_cols: list, flat_to_strings):
    """got_cols: list aligned with the view's projection (arrays, masked arrays or (sizes, bytes))."""
    exp = case["expect_cols"]
    if exp is None:
        return
    out_names = [n for n, _ in case["proj"]] if case["proj"] is not None else names
    assert len(out_names) == len(got_cols)
    for n, g in zip(out_names, got_cols):
        if n not in exp:
            continue
        want = exp[n]
        if isinstance(g, tuple):
            assert flat_to_strings(*g) == want, f"{case['name']}: column {n}"
        elif isinstance(g, np.ma.MaskedArray):
            assert [None if m else int(v) for v, m in zip(g.data.tolist(), np.ma.getmaskarray(g).tolist())] == want, f"{case['name']}: column {n}"
        else:
            assert g.tolist() == want, f"{case['name']}: column {n}: {g[:5]} vs {want[:5]}"
