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


def save_df(df, save_name):
    """CSV with the reference's column encoding (arrays as bracketed, comma-separated strings without spaces)."""
    if not save_name.endswith(".csv"):
        save_name += ".csv"
    df = df.copy(deep=True)
    df = df.astype(object)

    def enc(a):
        return np.array2string(np.asarray(a), separator=",", max_line_width=999).replace(" ", "")

    for i in range(len(df)):
        df.at[i, "field_strength"] = enc(df.at[i, "field_strength"])
        df.at[i, "receiver"] = enc(df.at[i, "receiver"])
        if df.at[i, "path_type"] != "Impossible":
            df.at[i, "tx_angles"] = enc(df.at[i, "tx_angles"])
            df.at[i, "rx_angles"] = enc(df.at[i, "rx_angles"])
    path = save_name if os.path.isabs(save_name) else os.path.join(results_dir(), save_name)
    df.to_csv(path, index=False)
    return path


def load_df(df_path):
    import pandas as pd

    df = pd.read_csv(df_path).astype(object)
    for i in range(len(df)):
        df.at[i, "receiver"] = string_arr_to_ndarray(df.at[i, "receiver"])
        df.at[i, "field_strength"] = string_complex_arr_to_ndarray(df.at[i, "field_strength"])
        if df.at[i, "path_type"] != "Impossible":
            df.at[i, "tx_angles"] = string_arr_to_ndarray(df.at[i, "tx_angles"])
            df.at[i, "rx_angles"] = string_arr_to_ndarray(df.at[i, "rx_angles"])
    return df
