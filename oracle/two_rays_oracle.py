"""TEST INFRASTRUCTURE ONLY (CPU; the product never imports this).

The reference's analytic two-ray model (electromagnetism_fun/theoretical_validation.py:31-170, flat ground), the known
answer its authors compare the simulator with (:246-275) on create_two_rays_place (place_utils.py:106-123): one line
of sight and one ground reflection between two antennas over a concrete plane.  Closed form, independent of the path
solver, the obstruction tests and the field code: it pins line of sight + single reflection + received power of the
whole path end to end."""
import numpy as np

from oracle import em_oracle as eo


def _unit(v):
    return v / np.linalg.norm(v)


def two_rays_geometry(L, ztx, zrx):
    """theoretical_validation.py:32-51: lengths and angles of the two rays (L = distance between the antenna bases)."""
    dlos = np.sqrt((ztx - zrx) ** 2 + L ** 2)
    dref = np.sqrt((ztx + zrx) ** 2 + L ** 2)
    d1 = ztx * dref / (ztx + zrx)
    L1 = L * ztx / (ztx + zrx)
    theta = np.arctan2(ztx + zrx, L)
    theta_tx = np.arcsin(2 * zrx * L / (dlos * dref))
    theta_rx = 2 * theta - theta_tx
    assert 0 < theta_rx < np.pi / 2 and 0 < theta_tx < np.pi / 2, "antennas are too close"
    return dlos, dref, d1, dref - d1, L1, L - L1, theta, theta_tx, theta_rx


def two_rays_powers(L, ztx, zrx, frequency=12.5e9):
    """theoretical_validation.py:106-167: (line-of-sight power, reflected power) in watts for P_IN = 1 W."""
    lam = eo.c / frequency
    K, Z0 = 2 * np.pi / lam, 120 * np.pi
    eps = eo.material_properties("concrete", frequency)[0]
    dlos, dref, d1, d2, L1, L2, theta, theta_tx, theta_rx = two_rays_geometry(L, ztx, zrx)
    ti = np.pi / 2 - theta                                   # :119
    root, ci = np.sqrt(eps - np.sin(ti) ** 2), np.cos(ti)    # :81-87, magnitudes only
    g_par = abs((eps * ci - root) / (eps * ci + root))
    g_per = abs((ci - root) / (ci + root))
    tx, rx, g = np.array([0.0, 0.0, ztx]), np.array([0.0, L, zrx]), np.array([0.0, L1, 0.0])
    dlos_v, d1_v, d2_v = _unit(rx - tx), _unit(g - tx), _unit(rx - g)
    x = np.array([1.0, 0.0, 0.0])
    rx_z = -dlos_v
    rx_basis = np.array([x, -_unit(np.cross(x, rx_z)), rx_z])  # :137-140
    pattern = eo.FieldOracle.radiation_pattern
    E0 = -1j * K * Z0 / (4 * np.pi) * x                      # :147, polarisation = world x
    Elos = E0 * pattern(0) * np.exp(-1j * K * dlos) / dlos
    per = _unit(np.cross(d1_v, d2_v))
    par = _unit(np.cross(per, d1_v))
    Eref = E0 * pattern(theta_tx) * np.exp(-1j * K * dref) / dref * (g_par * (x @ par) * par + g_per * (x @ per) * per)
    Elos, Eref = Elos @ rx_basis.T, Eref @ rx_basis.T        # world -> receiver frame (:158-159, :92-104)

    def effective_area(th):                                  # electromagnetism.py:144-148
        return lam ** 2 / (4 * np.pi) * eo.RX_GAIN * pattern(th)

    p_los = effective_area(0) / (2 * Z0) * np.linalg.norm(Elos.real) ** 2
    p_ref = effective_area(theta_rx) / (2 * Z0) * np.linalg.norm(Eref.real) ** 2
    p_ref *= abs(x @ (Eref / np.linalg.norm(Eref))) ** 2     # :164, the receiver's polarisation is its x axis
    assert np.linalg.norm(Eref) < np.linalg.norm(Elos)       # :169
    return p_los * eo.TX_GAIN, p_ref * eo.TX_GAIN


def two_rays_db(L, ztx, zrx, frequency=12.5e9):
    p_los, p_ref = two_rays_powers(L, ztx, zrx, frequency)
    return float(eo.to_db(p_los + p_ref))
