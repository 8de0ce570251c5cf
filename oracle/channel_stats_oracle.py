"""CPU restatement of the reference's channel statistics — TEST INFRASTRUCTURE ONLY (checker of the GPU path; never
imported by the product).  Follows electromagnetism_fun/electromagnetism_plots.py: compute_angular_windows :287-387,
compute_RMS_delay_spreads :389-400, compute_RMS_angular_spreads :403-431, dataframe_drop_NLOS :434-443,
compute_percent_coming_from_n_rays :445-460; angle_distance / to_cartesian electromagnetism_utils.py:38-42, 164-168.
Pinned by tests/golden/channel_stats.npz, which the reference's own functions produced (make_golden_stats.py).

Inputs are the table's columns as flat arrays: rx_id, possible (row has a path), is_los, rx_el, rx_az (degrees),
time, power."""
import numpy as np


def angle_distance(alpha, beta):
    phi = abs(beta - alpha) % 360
    return 360 - phi if phi > 180 else phi


def _receivers(rx_id, possible):
    ids = rx_id[possible]
    _, first = np.unique(ids, return_index=True)
    return ids[np.sort(first)]


def rms_delay_spreads(rx_id, possible, time, power):
    out = []
    for r in _receivers(rx_id, possible):
        m = possible & (rx_id == r)
        t, p = time[m], power[m]
        total = p.sum()
        t0 = t.min()
        t_mean = np.dot(t, p) / total - t0
        out.append(np.sqrt(np.dot((t - t0 - t_mean) ** 2, p) / total))
    return np.array(out)


def rms_angular_spreads(rx_id, possible, rx_el, rx_az, power):
    az_out, el_out = [], []
    for r in _receivers(rx_id, possible):
        m = possible & (rx_id == r)
        p, el, az = power[m], rx_el[m], rx_az[m]
        total = p.sum()
        top = int(np.argmax(p))
        d_el = np.deg2rad(np.array([angle_distance(x, el[top]) for x in el]))
        d_az = np.deg2rad(np.array([angle_distance(x, az[top]) for x in az]))
        el_mean, az_mean = np.dot(d_el, p) / total, np.dot(d_az, p) / total
        el_out.append(np.sqrt(np.dot((d_el - el_mean) ** 2, p) / total))
        az_out.append(np.sqrt(np.dot((d_az - az_mean) ** 2, p) / total))
    return np.degrees(np.array(az_out)), np.degrees(np.array(el_out))


def percent_coming_from_n_rays(rx_id, possible, is_los, power, n=1):
    out = []
    for r in _receivers(rx_id, possible):
        m = possible & (rx_id == r)
        if not is_los[m].any():
            continue  # dataframe_drop_NLOS
        p = np.sort(power[m])
        out.append(sum(p[-n:]) / power[m].sum() if len(p) >= n else sum(p) / power[m].sum())
    return np.array(out) * 100


def _cluster(power, total, centroid, points, power_percent):
    distances = np.sqrt(np.sum((points - centroid) ** 2, axis=1))
    members, gathered = [], 0
    for _ in range(len(distances)):
        if gathered / total >= power_percent:
            return members
        closest = int(np.argmin(distances))
        members.append(closest)
        distances[closest] = np.inf
        gathered += power[closest]
    return members


def _two_furthest(points):
    if len(points) == 2:
        return [0, 1]
    furthest, pair = 0, []
    for i, point in enumerate(points):
        distances = np.sqrt(np.sum((points - point) ** 2, axis=1))
        if np.max(distances) > furthest:
            furthest = np.max(distances)
            pair = [i, int(np.argmax(distances))]
    return pair


def angular_windows(rx_id, possible, rx_el, rx_az, power, power_percent=0.99):
    az_out, el_out = [], []
    for r in _receivers(rx_id, possible):
        m = possible & (rx_id == r)
        p, el, az = power[m], rx_el[m], rx_az[m]
        total = p.sum()
        cart = np.column_stack((1 * np.sin(el) * np.cos(az), 1 * np.sin(el) * np.sin(az), 1 * np.cos(el)))
        top = int(np.argmax(p))
        share = p[top] / total
        best_el = best_az = 0
        if share > power_percent:
            pass
        elif share > (1 - power_percent):
            i, j = _two_furthest(cart[_cluster(p, total, cart[top], cart, power_percent)])
            # the pair indexes the bundle, the reference reads the receiver's own list with it (:367-368)
            best_el, best_az = angle_distance(el[i], el[j]), angle_distance(az[i], az[j])
        else:
            min_spread = np.inf  # never lowered (:382-383): every centre replaces the result
            for c in cart:
                i, j = _two_furthest(cart[_cluster(p, total, c, cart, power_percent)])
                el_s, az_s = angle_distance(el[i], el[j]), angle_distance(az[i], az[j])
                if az_s * el_s < min_spread:
                    best_el, best_az = el_s, az_s
        el_out.append(best_el)
        az_out.append(best_az)
    return np.array(az_out), np.array(el_out)
