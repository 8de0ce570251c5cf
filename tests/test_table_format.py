"""The library's formatter of the result table's array columns (host code, no GPU) against NumPy's array printer, whose
strings the reference's CSV holds (electromagnetism_utils.py:104-117), and against the row-by-row Python formatter."""
import numpy as np


def _numpy_string(row):
    return np.array2string(row, separator=",", max_line_width=999).replace(" ", "")


def _random(rng, shape):
    v = rng.normal(size=shape) * 10.0 ** rng.integers(-12, 10, size=shape)
    v[rng.random(shape) < 0.05] = 0.0
    m = rng.random(shape) < 0.05
    v[m] = np.round(v[m])
    return v


def test_rows_are_formatted_as_numpy_prints_them():
    from raytracing_amd import _lib
    from raytracing_amd import electromagnetism_utils as emu

    rng = np.random.default_rng(0)
    big = rng.uniform(1e6, 1e8, (3000, 3)) * rng.choice([-1, 1], (3000, 3))  # 16 significant digits and more
    cases = [_random(rng, (4000, 3)), big, np.round(big, 3), rng.uniform(-180, 180, (4000, 2)),
             np.round(rng.uniform(-180, 180, (2000, 2)), 1), _random(rng, (3000, 3)) + 1j * _random(rng, (3000, 3)),
             (rng.normal(size=(3000, 3)) + 1j * rng.normal(size=(3000, 3))) * 1e-5,
             np.array([[0., 0, 0], [1, 2, 3], [1e8, 1, 1], [-0.0, 1e-5, 2], [123456789.123, 1, 2], [1e-300, 1e300, 1.],
                       [0.5, 0.25, 0.125], [1e-4, 1e-4, 0.1], [99999999., 1e5, 1e5]])]
    for a in cases:
        got = _lib.format_array_rows(a)
        assert got == [_numpy_string(r) for r in a]
        assert got == [emu.array_to_string(r) for r in a]
    # a row with a non-finite value is left to the caller
    a = np.array([[1.0, 2.0], [np.nan, 1.0], [3.0, np.inf]])
    assert _lib.format_array_rows(a) == ["[1.,2.]", None, None]
    assert _lib.format_array_rows(np.zeros((0, 3))) == []


def test_csv_is_the_one_the_row_by_row_formatter_writes(tmp_path, monkeypatch):
    from raytracing_amd import electromagnetism as em
    from raytracing_amd import electromagnetism_utils as emu

    monkeypatch.setenv("RAYTRACING_AMD_RESULTS", str(tmp_path))
    rng = np.random.default_rng(1)
    n, n_rx = 3000, 500
    rx = np.sort(rng.integers(0, n_rx - 20, n))
    df = em.table_from_columns(n_rx, rng.normal(size=(n_rx, 3)) * 100, rx, em.path_labels(rng.choice([0, 1, 2, 256, 257], n)),
                               rng.uniform(-180, 180, (n, 4)), rng.random(n) * 1e-6,
                               (rng.normal(size=(n, 3)) + 1j * rng.normal(size=(n, 3))) * 1e-5, rng.random(n) * 1e-9)
    assert (df["path_type"] == "Impossible").sum() >= 20
    path = emu.save_df(df, "native")
    out = df.copy(deep=False).astype(object)
    possible = [t != "Impossible" for t in df["path_type"]]
    out["field_strength"] = [emu.array_to_string(v) for v in df["field_strength"]]
    out["receiver"] = [emu.array_to_string(v) for v in df["receiver"]]
    for col in ("tx_angles", "rx_angles"):
        out[col] = [emu.array_to_string(v) if ok else v for v, ok in zip(df[col], possible)]
    out.to_csv(tmp_path / "rowwise.csv", index=False)
    assert open(path).read() == open(tmp_path / "rowwise.csv").read()
    back = emu.load_df(path)
    assert list(back["path_type"]) == list(df["path_type"])
