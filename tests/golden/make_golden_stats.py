"""Generates tests/golden/channel_stats.npz from the REFERENCE's channel statistics
(electromagnetism_fun/electromagnetism_plots.py:287-460; authoring container only, see make_golden.py).

The module also draws figures and imports seaborn for that; seaborn is not installed and contributes nothing to the
four statistics, so an empty module stands under its name.  The statistics themselves (pandas / NumPy arithmetic)
run untouched on seeded synthetic result tables of the shape my_field_computation writes.
"""
import os
import sys
import types
import warnings

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
warnings.filterwarnings("ignore")

from _ref_import import import_reference_electromagnetism  # noqa: E402

cwd = os.getcwd()
geom, em = import_reference_electromagnetism()
sys.modules.setdefault("seaborn", types.ModuleType("seaborn"))
import matplotlib  # noqa: E402

matplotlib.use("Agg")
import electromagnetism_fun.electromagnetism_plots as plots  # noqa: E402

os.chdir(cwd)


def table(rng, n_rx, max_paths, dominant):
    """Result table with 1..max_paths paths per receiver (some receivers without any path, some without LOS)."""
    rows = []
    for r in range(n_rx):
        n = int(rng.integers(0, max_paths + 1))
        if n == 0:
            rows.append({"rx_id": r, "path_type": "Impossible", "tx_angles": None, "rx_angles": None,
                         "time_to_receiver": None, "path_power": 0.0})
            continue
        power = 10.0 ** rng.uniform(-12, -7, size=n)
        if dominant and rng.random() < 0.5:
            power[int(rng.integers(0, n))] *= 10.0 ** rng.uniform(1, 4)
        types_ = ["LOS" if (k == 0 and rng.random() < 0.7) else str(rng.choice(["R", "RR", "D", "RD"])) for k in range(n)]
        for k in range(n):
            rows.append({"rx_id": r, "path_type": types_[k],
                         "tx_angles": np.array([rng.uniform(0, 180), rng.uniform(-180, 180)]),
                         "rx_angles": np.array([rng.uniform(0, 180), rng.uniform(-180, 180)]),
                         "time_to_receiver": float(rng.uniform(1e-8, 2e-6)), "path_power": float(power[k])})
    return pd.DataFrame(rows)


def flat(df):
    """Arrays of the table (NaN where a row has no path)."""
    ok = df["path_type"].values != "Impossible"
    n = len(df)
    el = np.full(n, np.nan)
    az = np.full(n, np.nan)
    el[ok] = [a[0] for a in df["rx_angles"][ok]]
    az[ok] = [a[1] for a in df["rx_angles"][ok]]
    t = np.array([np.nan if v is None else v for v in df["time_to_receiver"]], dtype=float)
    return {"rx_id": df["rx_id"].values.astype(np.int64), "is_los": (df["path_type"].values == "LOS"),
            "possible": ok, "rx_el": el, "rx_az": az, "time": t, "power": df["path_power"].values.astype(float)}


def main():
    rng = np.random.default_rng(2024)
    out = {}
    cases = [("a", 40, 12, True), ("b", 25, 40, False), ("c", 30, 3, True), ("d", 12, 90, True)]
    for name, n_rx, max_paths, dominant in cases:
        df = table(rng, n_rx, max_paths, dominant)
        for k, v in flat(df).items():
            out[f"{name}_{k}"] = v
        out[f"{name}_delay"] = plots.compute_RMS_delay_spreads(df)
        az, el = plots.compute_RMS_angular_spreads(df)
        out[f"{name}_rms_az"], out[f"{name}_rms_el"] = az, el
        for n in (1, 2, 5):
            out[f"{name}_percent_{n}"] = plots.compute_percent_coming_from_n_rays(df, n)
        for pp in (0.99, 0.7):
            az, el = plots.compute_angular_windows(df, POWER_PERCENT=pp)
            out[f"{name}_win_az_{int(pp * 100)}"], out[f"{name}_win_el_{int(pp * 100)}"] = az, el
    np.savez_compressed(os.path.join(HERE, "channel_stats.npz"), **out)
    print({k: v.shape for k, v in out.items() if k.startswith("a_")})


if __name__ == "__main__":
    main()
