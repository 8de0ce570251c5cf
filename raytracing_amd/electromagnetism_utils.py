"""Helpers of the field step with the reference's names (``electromagnetism_fun/electromagnetism_utils.py``:
to_spherical :146-161, cot / vv_normalize :170-177, to_db :180-188, get_power_db_each_receiver :104-115,
save_df / load_df :217-269, path-type helpers :48-65)."""
import os

import numpy as np

from .materials_properties import P_IN


def generate_path_types_of_order(order):
    if order == 0:
        return ["LOS"]
    return ["R" * order, "R" * (order - 1) + "D"]


def generate_path_types_up_to_order(order):
    out = []
    for i in range(order + 1):
        out.extend(generate_path_types_of_order(i))
    return out


def to_spherical(point):
    x, y, z = point
    assert not np.all(np.asarray(point) == 0), "spherical coordinates system not defined for point at the origin."
    if x == 0 and y == 0:
        return z, 0, 0
    hxy = np.sqrt(x ** 2 + y ** 2)
    return np.sqrt(hxy ** 2 + z ** 2), np.arctan2(hxy, z), np.arctan2(y, x)


def cot(x):
    return 1 / np.tan(x)


def vv_normalize(vv):
    norm = np.linalg.norm(vv)
    if norm == 0:
        return np.array([0, 0, 0])
    return vv / norm


def to_db(field_power):
    """Power in watts to dB relative to P_IN; zero power gives -inf."""
    P = np.atleast_1d(field_power).astype(float)
    db = np.full(P.shape, -np.inf)
    ind = P != 0
    db[ind] = 10 * np.log10(P[ind] / P_IN)
    return db[0] if np.isscalar(field_power) else db


def get_power_db_each_receiver(df):
    """Total power of each receiver in dB (sum of its paths' powers)."""
    power_sum = df.groupby("rx_id")["path_power"].sum().values
    return to_db(power_sum)


def get_receiver_coordinates(df, two_dimensional=False):
    coords = np.array(df.groupby("rx_id")["receiver"].first().tolist())
    return coords[:, :2] if two_dimensional else coords


def string_arr_to_ndarray(string_arr):
    if not isinstance(string_arr, np.ndarray):
        string_arr = np.fromstring(string_arr[1:-1], sep=",")
    return string_arr


def string_complex_arr_to_ndarray(string):
    split = string[1:-1].split(",")
    assert len(split) == 3, f"strange array shape: split: {split}"
    return np.array([complex(c) for c in split])


def results_dir():
    """Where the CSV / JSON results go: the reference writes to ``../results`` relative to its module directory;
    here it is ``./results`` unless RAYTRACING_AMD_RESULTS names another directory."""
    d = os.environ.get("RAYTRACING_AMD_RESULTS", os.path.join(os.getcwd(), "results"))
    os.makedirs(d, exist_ok=True)
    return d


def _float_strings(values):
    """Strings NumPy's array printer gives the elements of a 1-D float array (default print options, spaces
    removed), or None when the array holds a non-finite value (the caller then asks NumPy itself)."""
    vals = [float(v) for v in values]
    if not all(np.isfinite(vals)):
        return None
    nonzero = [abs(v) for v in vals if v != 0.0]
    exp_format = bool(nonzero) and (max(nonzero) >= 1e8 or min(nonzero) < 1e-4 or max(nonzero) / min(nonzero) > 1e3)
    if not exp_format:
        return [np.format_float_positional(v, precision=8, unique=True, fractional=True, trim=".") for v in vals]
    first = [np.format_float_scientific(v, precision=8, unique=True, trim=".") for v in vals]
    mantissas, exponents = zip(*(t.split("e") for t in first))
    digits = max(len(m.split(".")[1]) for m in mantissas)
    exp_digits = max(len(e) for e in exponents) - 1
    return [np.format_float_scientific(v, precision=digits, unique=True, min_digits=digits, trim="k", exp_digits=exp_digits)
            for v in vals]


def array_to_string(a):
    """``np.array2string(a, separator=",").replace(" ", "")`` of a short 1-D real or complex array — the encoding of
    the array columns of the reference's CSV (electromagnetism_utils.py:104-117) — without the array printer's set-up
    cost per call; anything else goes through NumPy."""
    a = np.asarray(a)
    if a.ndim == 1 and 0 < a.size <= 16:
        if a.dtype.kind == "f":
            parts = _float_strings(a)
            if parts is not None:
                return "[" + ",".join(parts) + "]"
        elif a.dtype.kind == "c":
            re, im = _float_strings(a.real), _float_strings(a.imag)
            if re is not None and im is not None:
                return "[" + ",".join(r + (i if i[0] == "-" else "+" + i) + "j" for r, i in zip(re, im)) + "]"
    return np.array2string(a, separator=",", max_line_width=999).replace(" ", "")


def _format_rows(block):
    from . import _lib

    return _lib.format_array_rows(block)


def _is_missing(v):
    return isinstance(v, float) and np.isnan(v)


def save_df(df, save_name):
    """CSV with the reference's column encoding (arrays as bracketed, comma-separated strings without spaces)."""
    if not save_name.endswith(".csv"):
        save_name += ".csv"
    out = df.copy(deep=False).astype(object)
    possible = [t != "Impossible" for t in df["path_type"]]
    rows_with_path = [i for i, ok in enumerate(possible) if ok]

    def array_column(values, fallback):
        """The column's strings: the rows that hold a path through the library's formatter, all at once (the same
        strings as ``array_to_string``, which the other rows and anything unusual still go through)."""
        values = list(values)
        out_col = [fallback(v) for v in values] if not rows_with_path else [None] * len(values)
        if rows_with_path:
            try:
                block = np.array([values[i] for i in rows_with_path])
                strings = _format_rows(block) if block.ndim == 2 and block.dtype.kind in "fc" else None
            except (ValueError, TypeError):
                strings = None
            done = set()
            if strings is not None:
                for i, text in zip(rows_with_path, strings):
                    if text is not None:
                        out_col[i] = text
                        done.add(i)
            for i in range(len(values)):
                if i not in done:
                    out_col[i] = fallback(values[i])
        return out_col

    out["field_strength"] = array_column(df["field_strength"], array_to_string)
    seen = {}  # a receiver has many rows: its position is formatted once

    def receiver_string(v):
        key = np.asarray(v, dtype=float).tobytes()
        if key not in seen:
            seen[key] = array_to_string(v)
        return seen[key]

    out["receiver"] = [receiver_string(v) for v in df["receiver"]]
    for col in ("tx_angles", "rx_angles"):
        out[col] = array_column(df[col], lambda v: array_to_string(v) if v is not None and not _is_missing(v) else v)
    path = save_name if os.path.isabs(save_name) else os.path.join(results_dir(), save_name)
    out.to_csv(path, index=False)
    return path


def load_df(df_path):
    import pandas as pd

    df = pd.read_csv(df_path).astype(object)
    possible = [t != "Impossible" for t in df["path_type"]]
    df["receiver"] = [string_arr_to_ndarray(v) for v in df["receiver"]]
    df["field_strength"] = [string_complex_arr_to_ndarray(v) for v in df["field_strength"]]
    for col in ("tx_angles", "rx_angles"):
        df[col] = [string_arr_to_ndarray(v) if ok else v for v, ok in zip(df[col], possible)]
    return df
