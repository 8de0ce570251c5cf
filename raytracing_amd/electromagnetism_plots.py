"""Channel statistics of the result table with the reference's names
(``electromagnetism_fun/electromagnetism_plots.py``: compute_angular_windows :287-387, compute_RMS_delay_spreads
:389-400, compute_RMS_angular_spreads :403-431, dataframe_drop_NLOS :434-443, compute_percent_coming_from_n_rays
:445-460).  The figures of that module are not part of this package; the statistics are computed on the GPU, one
wave per receiver (``rt_channel_stats``)."""
import numpy as np

from . import _lib


def _grouped(df):
    """Rows that have a path, grouped by receiver in order of first appearance (``df['rx_id'].unique()``), row order
    kept inside a group: (receiver ids, offsets, arrays per row)."""
    df = df.dropna(subset=["tx_angles", "rx_angles"])
    rx_id = df["rx_id"].to_numpy()
    receivers, first = np.unique(rx_id, return_index=True)
    receivers = receivers[np.argsort(first, kind="stable")]
    rank = {r: k for k, r in enumerate(receivers.tolist())}
    order = np.argsort(np.array([rank[r] for r in rx_id.tolist()], dtype=np.int64), kind="stable")
    counts = np.bincount(np.array([rank[r] for r in rx_id.tolist()], dtype=np.int64), minlength=len(receivers))
    off = np.zeros(len(receivers) + 1, dtype=np.int64)
    np.cumsum(counts, out=off[1:])
    angles = np.array([np.asarray(a, dtype=float) for a in df["rx_angles"].to_numpy()[order]], dtype=float).reshape(-1, 2)
    el, az = angles[:, 0].copy(), angles[:, 1].copy()
    # the point of the unit sphere as the reference forms it: to_cartesian(r=1, theta=rx_el, phi=rx_az) on the
    # table's angles as they are (electromagnetism_plots.py:353, electromagnetism_utils.py:164-168)
    xyz = np.column_stack((1 * np.sin(el) * np.cos(az), 1 * np.sin(el) * np.sin(az), 1 * np.cos(el)))
    return receivers, off, {
        "power": df["path_power"].to_numpy(dtype=float)[order],
        "time": df["time_to_receiver"].to_numpy(dtype=float)[order],
        "el": el, "az": az, "xyz": xyz,
        "is_los": (df["path_type"].to_numpy()[order] == "LOS"),
    }


def channel_statistics(df, n=1, POWER_PERCENT=0.99):
    """Every statistic of this module in one launch: dict with ``receivers`` (ids in the reference's order) and one
    array per statistic (``delay``, ``rms_az``, ``rms_el``, ``percent``, ``has_los``, ``window_az``, ``window_el``)."""
    receivers, off, a = _grouped(df)
    out = _lib.channel_stats(off, a["power"], a["time"], a["el"], a["az"], a["xyz"], a["is_los"], n, POWER_PERCENT)
    return {"receivers": receivers, "delay": out[:, 0], "rms_az": out[:, 1], "rms_el": out[:, 2], "percent": out[:, 3],
            "has_los": out[:, 4] > 0, "window_az": out[:, 5], "window_el": out[:, 6]}


def compute_RMS_delay_spreads(df):
    return channel_statistics(df)["delay"]


def compute_RMS_angular_spreads(df):
    s = channel_statistics(df)
    return s["rms_az"], s["rms_el"]


def dataframe_drop_NLOS(dataframe):
    """Rows of the receivers that have a line of sight."""
    df = dataframe.dropna(subset=["tx_angles", "rx_angles"])
    with_los = set(df.loc[df["path_type"] == "LOS", "rx_id"].tolist())
    return df[df["rx_id"].isin(with_los)].copy()


def compute_percent_coming_from_n_rays(df, n=1):
    s = channel_statistics(df, n=n)
    return s["percent"][s["has_los"]]


def compute_angular_windows(df, POWER_PERCENT=0.99):
    s = channel_statistics(df, POWER_PERCENT=POWER_PERCENT)
    if np.isnan(s["window_az"]).any():
        # a bundle of a single ray: the reference fails unpacking the two furthest rays (:367)
        raise ValueError("not enough values to unpack (expected 2, got 0)")
    return s["window_az"], s["window_el"]
