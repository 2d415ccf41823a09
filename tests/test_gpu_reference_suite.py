"""GPU: the reference's own integration tests re-expressed on the Python mirror of its API
(test/view.jl, test/columnbroadcast.jl, test/column.jl, test/rows.jl) with pandas standing in for
DataFrames.jl.  Rows are 1-based and ranges inclusive, as in Julia: jr(a, b) == a:b, END == end."""
import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu

SZ = 1000


@pytest.fixture(scope="module")
def env(oracle, dfdb_mod, ctx, tmp_path_factory):
    """df + tb = create_table("test_data", from = df; block_size = 100): written by the oracle's writer in the
    reference format, opened (LZ4-decoded on the device) by the engine."""
    a = np.arange(1, SZ + 1, dtype=np.int64)
    b = [str(i) for i in range(1, SZ + 1)]
    df = pd.DataFrame({"a": a, "b": np.array(b, dtype=object), "c": a.copy()})
    t = oracle.Table(block_size=100)
    t.add_column("a", a); t.add_column("b", b); t.add_column("c", a)
    path = str(tmp_path_factory.mktemp("ref") / "test_data")
    t.save(path)
    tb = dfdb_mod.open_table(path)
    return dfdb_mod, df, tb


def eq(dft: pd.DataFrame, want: pd.DataFrame) -> bool:
    want = want.reset_index(drop=True)
    if list(dft.columns) != list(want.columns) or len(dft) != len(want):
        return False
    return all(np.array_equal(dft[c].to_numpy(), want[c].to_numpy()) for c in want.columns)


def test_view(env):                                     # test/view.jl:6-141
    D, df, tb = env
    ALL, END, jr = D.ALL, D.END, D.jr
    materialize, selection, projection, selproj = D.materialize, D.selection, D.projection, D.selproj
    assert tb.names() == list(df.columns)
    v1 = D.DFView(tb)
    v2 = selection(v1, jr(1, 1000))
    v3 = selection(v2, (("a",), lambda a: a % 50 == 0))
    v4 = selection(v3, ("c", lambda a: a < 930))
    v4 = projection(v4, {"a": (("a",), lambda a: a / 50)})

    assert eq(materialize(v1), df) and D.nrow(v1) == len(df)
    assert eq(materialize(v2), df) and D.nrow(v2) == len(df)
    assert D.size(v2) == df.shape and D.size(v2, 1) == df.shape[0] and D.size(v2, 2) == df.shape[1]
    dft = materialize(v3)
    assert eq(dft, df[df.a % 50 == 0]) and D.nrow(v3) == len(dft)
    dft = materialize(v4)
    ind = (df.a % 50 == 0) & (df.c < 930)
    assert eq(dft, pd.DataFrame({"a": df.a[ind] / 50})) and D.nrow(v4) == len(dft) and D.size(v4) == dft.shape
    with pytest.raises(ValueError):                     # @test_throws ArgumentError projection(v4, (c=:c,))
        projection(v4, {"c": "c"})

    tv = projection(v1, ["a", "c"]);                    assert D.size(tv, 2) == 2 and eq(materialize(tv), df[["a", "c"]])
    tv = projection(v1, [("a", "a"), ("c", ("c", lambda c: c * 2))])
    assert D.size(tv, 2) == 2 and eq(materialize(tv), pd.DataFrame({"a": df.a, "c": df.c * 2}))
    tv = projection(v1, [1, 3]);                        assert eq(materialize(tv), df[["a", "c"]])
    tv = projection(v1, jr(1, 2));                      assert eq(materialize(tv), df[["a", "b"]])
    tv = selproj(v1, ("a", lambda a: a % 50 == 0), ["c"])
    assert D.size(tv, 2) == 1 and eq(materialize(tv), df[df.a % 50 == 0][["c"]])
    tv = selproj(v1, 1, ["c"]);                         assert eq(materialize(tv), df.iloc[[0]][["c"]])
    tv = selproj(v1, [1, 200], ["c"]);                  assert eq(materialize(tv), df.iloc[[0, 199]][["c"]])
    tv = v1[("a", lambda a: a % 50 == 0), ["c"]];       assert eq(materialize(tv), df[df.a % 50 == 0][["c"]])
    tv = v1[[1, 200], ["c"]];                           assert eq(materialize(tv), df.iloc[[0, 199]][["c"]])
    tv = v1[jr(1, 200), ALL]
    assert D.size(tv, 2) == 3 and D.size(tv, 1) == 200 and eq(materialize(tv), df.iloc[0:200])
    tv = v1[ALL, {"e": "a"}]
    assert D.size(tv) == (1000, 1) and eq(materialize(tv), pd.DataFrame({"e": df.a}))
    assert v1[ALL, ALL] is v1
    tv = tb[ALL, {"e": "a"}];                           assert D.size(tv) == (1000, 1)
    tv = tb[jr(END - 10, END), {"e": "a"}]
    assert D.size(tv) == (11, 1) and eq(materialize(tv), pd.DataFrame({"e": df.a.iloc[-11:]}))
    tv = tb[jr(END - 10, END), jr(END - 1, END)]
    assert D.size(tv) == (11, 2) and eq(materialize(tv), df.iloc[-11:, -2:])

    assert tb[jr(1, 20), ["a", "b"]] == tb[jr(1, 20), ["a", "b"]]
    assert tb[10, "a"] == df.a.iloc[9]
    assert tb[jr(1, 30), ["a", "b"]] != tb[jr(1, 20), ["a", "b"]]
    assert not D.issameselection(tb[jr(1, 30), ["a", "b"]], tb[jr(1, 20), ["a", "b"]])
    assert tb[jr(1, 20), ["a", "b"]] != tb[jr(1, 20), ["b", "a"]]
    tff = lambda a: a % 50 == 0
    tv = selproj(v1, ("a", tff), ["c"])
    tv2 = tb[("a", tff), ALL]
    assert tv != tv2 and D.issameselection(tv, tv2) and tv == tv2[ALL, ["c"]]
    # two distinct closure objects give unequal views (quirk Q12)
    assert tb[("a", lambda a: a % 50 == 0), ALL] != tb[("a", lambda a: a % 50 == 0), ALL]


def test_column_broadcast(env):                         # test/columnbroadcast.jl:7-63
    D, df, tb = env
    jr, materialize = D.jr, D.materialize
    with pytest.raises(ValueError):                     # different selections
        tb.a[jr(1, 20)] + tb.c[jr(11, 30)]
    r = tb.a[jr(1, 20)] + 20
    assert np.array_equal(materialize(r), df.a[:20] + 20)
    assert np.array_equal(materialize(tb.a[jr(1, 20)] * tb.a[jr(1, 20)]), df.a[:20] * df.a[:20])
    assert np.array_equal(materialize(tb.a[jr(1, 20)] * tb.a[jr(1, 20)] - 20), df.a[:20] * df.a[:20] - 20)
    assert np.array_equal(materialize(tb.a * tb.c), df.a * df.c)
    assert np.array_equal(materialize(tb.a == 10), df.a == 10)
    test_t = np.empty(SZ, np.int64)
    (tb.a * tb.c).copyto(test_t);                       assert np.array_equal(test_t, df.a * df.c)
    tb.a.copyto(test_t);                                assert np.array_equal(test_t, df.a)
    test2 = (300 >= tb.a) & (tb.a >= 10)                # 300 .>= tb.a .>= 10
    tb2 = tb[test2, D.ALL]
    df2 = df[(300 >= df.a) & (df.a >= 10)]
    assert len(df2) == 291 and eq_df(materialize(tb2), df2)
    tb3 = tb2[D.startswith(tb2.b, "1"), D.ALL]
    df3 = df2[df2.b.str.startswith("1")]
    assert len(df3) == 110 and eq_df(materialize(tb3), df3)
    v = D.view_from_columns(a=tb.a * 3, g=tb.a * tb.c)
    assert eq_df(materialize(v), pd.DataFrame({"a": df.a * 3, "g": df.a * df.c}))
    with pytest.raises(TypeError):
        tb.a * df.a.to_numpy()                          # arrays are not column style (columnbroadcast.jl:16-17)


def eq_df(a, b):
    return eq(a, b)


def test_columns(env):                                  # test/column.jl:6-54
    D, df, tb = env
    jr, materialize, ALL = D.jr, D.materialize, D.ALL
    col = tb[ALL, 1]
    assert isinstance(col, D.DFColumn) and isinstance(tb[ALL, [1]], D.DFView) and isinstance(tb[ALL, ["a"]], D.DFView)
    assert isinstance(tb[ALL, "a"], D.DFColumn) and isinstance(tb[jr(1, 5, D.END), "a"], D.DFColumn)
    assert len(col) == SZ and np.array_equal(materialize(col), df.a)
    assert col.eltype == D.ir.I64
    assert np.array_equal(col.collect(), materialize(col)) and list(dict.fromkeys(col)) == list(col.collect())
    col2 = col[jr(90, 110)]
    assert np.array_equal(col2.collect(), df.a[89:110]) and col2[1] == 90 and col2[12] == 101
    with pytest.raises(IndexError):
        col2[22]
    col3 = tb[ALL, (("a", "c"), lambda a, c: a + c * 2)]
    assert np.array_equal(materialize(col3), df.a + df.c * 2)
    col3 = tb[ALL, ("a", lambda a: a * 4)]
    assert np.array_equal(materialize(col3), df.a * 4)
    assert D.col_equal(tb.a, tb[ALL, "a"]) and D.col_equal(tb[jr(1, 20), ALL].a, tb[jr(1, 20), "a"])
    # docs/src/index.md:503 style aggregate over a filtered column
    price = tb.c[tb.b == "77"]
    assert price.sum() == 77 and len(price) == 1 and price.mean() == 77.0


def test_rows_and_errors(env):                          # test/rows.jl:20-29, test/view.jl:54
    D, df, tb = env
    r = tb[3, D.ALL]
    assert r == {"a": 3, "b": "3", "c": 3}
    with pytest.raises(IndexError):
        tb[SZ + 1, D.ALL]
    with pytest.raises(AttributeError):
        tb.nope
    with pytest.raises(ValueError):                     # predicate must be Bool (selection.jl:52-55)
        tb[("a", lambda a: a * 3), D.ALL]
    with pytest.raises(ValueError):
        D.DFColumn(tb[D.ALL, ["a", "b"]])               # "Column projection must contains singe element"
    assert D.nrow(tb) == SZ and D.ncol(tb) == 3 and D.size(tb) == (SZ, 3)
    assert D.head(tb, 5)["a"].tolist() == [1, 2, 3, 4, 5]
    # selection over a computed projection column nests the tree (quirk Q14)
    v = tb[D.ALL, {"k": (("a", "c"), lambda a, c: a + c), "b": "b"}]
    v = v[("k", lambda k: k > 1990), D.ALL]
    assert D.materialize(v)["k"].tolist() == [1992, 1994, 1996, 1998, 2000]
    # (:c, :a) => f passes arguments in projection order (quirk Q13)
    v = tb[(("c", "a"), lambda x, y: (x == y) & (y < 3)), ["a"]]
    assert D.materialize(v)["a"].tolist() == [1, 2]


def test_create_from_data(oracle, dfdb_mod, ctx, tmp_path):     # test/create_from_data.jl:5-41
    """"from DataFrame": a = 1:5, b = string.(1:5) -> create_table(from = df) -> materialize == df, size == size(df).  "from rows": the rows of the
    reference's own test/test.csv (kept as tests/golden/reference_test.csv), block_size = 10, the table the test ends with — those rows four times
    (`repeat(df[:, n], 4)`; the reference gets there with three `insert`s, which rewrite the partial last block: this engine has no insert, the 28 rows
    are written at once, 3 blocks, the last partial).  Both tables are also read back by the oracle's reader (liblz4)."""
    import csv, os
    D = dfdb_mod
    df = pd.DataFrame({"a": np.arange(1, 6, dtype=np.int64), "b": np.array([str(i) for i in range(1, 6)], dtype=object)})
    tb = D.create_table(str(tmp_path / "test_data"), from_={"a": df.a.to_numpy(), "b": list(df.b)})
    assert eq(D.materialize(tb), df)
    assert D.size(tb) == df.shape and D.nrow(tb) == 5 and D.ncol(tb) == 2
    tb.close()
    with open(os.path.join(os.path.dirname(__file__), "golden", "reference_test.csv")) as f:
        rows = list(csv.DictReader(f))
    assert [r["a"] for r in rows] == [str(i) for i in range(1, 8)] and [r["b"] for r in rows] == [str(i) for i in range(1, 8)]
    four = {n: [r[n] for r in rows] * 4 for n in ("a", "b")}     # CSV.Rows yields strings for every column
    path = str(tmp_path / "test_rows")
    tb = D.create_table(path, from_=four, block_size=10)
    dft = D.materialize(tb)
    for n in ("a", "b"):
        assert list(dft[n]) == four[n]
    assert D.size(tb) == (28, 2)
    ot = oracle.Table.open(path)
    assert ot.block_size == 10 and [oracle.flat_to_strings(*c) for c in ot.view().materialize()] == [four["a"], four["b"]]
    tb.close()


def test_add_column(oracle, dfdb_mod, ctx, tmp_path):           # test/table_changes.jl:57-112 (the add_column! testset)
    """add_column!(tb, :e, (1:1000) .+ 2000) -> materialize(tb) has e after c; a lazy column of the SAME table and of ANOTHER table of the same shape
    (`tb[:, (c = :c => c -> c * 3,)][:, :c]`) -> materialize(tb[:, :e]) == df.c .* 3; a name that exists and a column of the wrong length are ArgumentErrors.
    (`before = :a` — where the new column sits in meta.bin — is the control plane's: not mirrored.)"""
    D = dfdb_mod
    sz = 1000
    a = np.arange(1, sz + 1, dtype=np.int64)
    df = pd.DataFrame({"a": a, "b": np.array([str(i) for i in a], dtype=object), "c": a.astype(np.int16)})
    cols = {"a": a, "b": list(df.b), "c": df.c.to_numpy()}
    tb = D.create_table(str(tmp_path / "test_data"), from_=cols, block_size=100)
    tb.load()
    with pytest.raises(ValueError):
        tb.add_column("c", np.zeros(sz, np.int64))        # the name exists
    with pytest.raises(ValueError):
        tb.add_column("e", np.zeros(0, np.int64))         # wrong length
    tb.add_column("e", a + 2000)
    want = df.copy(); want["e"] = a + 2000
    assert eq(D.materialize(tb), want)
    tb2 = D.create_table(str(tmp_path / "test_data2"), from_=cols, block_size=100)
    tb2.load()
    v1 = tb[D.ALL, {"c": ("c", lambda c: c * 3)}][D.ALL, "c"]
    tb.add_column_from("f", v1)
    assert np.array_equal(D.materialize(tb[D.ALL, "f"]), df.c.to_numpy() * 3)
    v2 = tb2[D.ALL, {"c": ("c", lambda c: c * 3)}][D.ALL, "c"]
    tb.add_column_from("g", v2)
    assert np.array_equal(D.materialize(tb[D.ALL, "g"]), df.c.to_numpy() * 3)
    tb.close(); tb2.close()


def test_type_strings_written_by_hand(oracle, dfdb_mod, ctx, tmp_path):   # test/column_types.jl:31-50
    """a table directory whose type strings were typed here as the Julia package writes them — "Int32", "Missing(Int32)" (deserialize: column_types.jl:31-38) —
    opens in the ENGINE as those types and decodes to the values put in; "Tuple(Int32, UInt64)" (:46-50) is refused by name, as the oracle refuses it"""
    import os, struct
    D = dfdb_mod
    hand = str(tmp_path / "hand"); os.mkdir(hand)

    def jstr(s_):
        return struct.pack("<i", len(s_)) + s_.encode()
    open(os.path.join(hand, "meta.bin"), "wb").write(struct.pack("<qqq", 1, 4, 2) + struct.pack("<q", 1) + jstr("x") + jstr("Int32") + struct.pack("<q", 2) + jstr("y") + jstr("Missing(Int32)"))
    open(os.path.join(hand, "1.bin"), "wb").write(struct.pack("<q", 4) + jstr("Int32") + oracle.block_encode(struct.pack("<ii", 7, -8), 2))
    open(os.path.join(hand, "2.bin"), "wb").write(struct.pack("<q", 4) + jstr("Missing(Int32)") + oracle.block_encode(struct.pack("<Q", 0b10) + struct.pack("<ii", 5, 99), 2))
    tb = D.open_table(hand)
    assert [c.dtype for c in tb.columns_meta()] == [D.ir.I32, D.ir.I32 | D.ir.NULLABLE]
    got = D.materialize(tb)
    assert got["x"].tolist() == [7, -8] and got["y"][0] == 5 and pd.isna(got["y"][1])
    tb.close()
    open(os.path.join(hand, "meta.bin"), "wb").write(struct.pack("<qqq", 1, 4, 1) + struct.pack("<q", 1) + jstr("x") + jstr("Tuple(Int32, UInt64)"))
    with pytest.raises(Exception, match="Tuple"):
        D.open_table(hand)
