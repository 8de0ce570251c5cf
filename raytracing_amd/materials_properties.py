"""Radio constants and ITU-R P.2040-2 material table of the field step.

Same names and default values as the reference's ``electromagnetism_fun/materials_properties.py``
(:19-41 constants, :47-104 materials, :107-109 effective permittivity, :112-143 ``set_properties``).
The reference fixes the carrier by editing the module; here ``material_table(frequency)`` gives the table at
any carrier so that a frequency sweep does not reload anything, and the module-level names hold the
12.5 GHz defaults.
"""
import numpy as np
from scipy.constants import c as C_LIGHT, epsilon_0 as EPSILON_0, pi  # the reference takes them from SciPy too (:13)

N_TREES = 10
MIN_TREE_HEIGHT = 10
MAX_TREE_HEIGHT = 20
TREE_SIZE = 2
FREQUENCY = 12.5 * 1e9

LAMBDA = C_LIGHT / FREQUENCY
K = 2 * pi / LAMBDA
Z_0 = 120 * pi

P_IN = 1
RADIATION_EFFICIENCY = 1
RADIATION_POWER = RADIATION_EFFICIENCY * P_IN

ISOTROPIC_ANTENNA = True
ALPHA = 10

TX_GAIN, RX_GAIN = 1, 1
ITU = "ITU-R P.2040-2"

MATERIAL_NAMES = ("air", "concrete", "brick", "wood", "glass", "metal", "medium_dry_ground")
MATERIAL_ID = {name: i for i, name in enumerate(MATERIAL_NAMES)}

_BUILDING_TYPE_TO_MATERIAL = {
    "office": "glass",
    "appartments": "brick",
    "garage": "concrete",
    "ground": "medium_dry_ground",
    "road": "concrete",
    "tree_trunk": "wood",
    "tree_crown": "wood",
}


def material_records(frequency=FREQUENCY):
    """List of material dicts at a carrier frequency in Hz (reference :47-101)."""
    f_ghz = frequency / 1e9
    return [
        dict(material="air", epsilon=1, mu=1, sigma=0, roughness=None,
             frequency_range_GHz="0.001-100", source=ITU),
        dict(material="concrete", epsilon=5.24, mu=1, sigma=0.0462 * f_ghz ** 0.7822, roughness=1e-4,
             frequency_range_GHz="1-100", source=ITU),
        dict(material="brick", epsilon=3.91, mu=1, sigma=0.0238 * f_ghz ** 0.16, roughness=1e-3,
             frequency_range_GHz="1-40", source=ITU),
        dict(material="wood", epsilon=1.99, mu=1, sigma=0.0047 * f_ghz ** 1.0718, roughness=1e-4,
             frequency_range_GHz="0.001-100", source=ITU),
        dict(material="glass", epsilon=6.31, mu=1, sigma=0.0036 * f_ghz ** 1.3394, roughness=1e-6,
             frequency_range_GHz="0.1-100", source=ITU),
        dict(material="metal", epsilon=1, mu=1, sigma=1e7, roughness=1e-6,
             frequency_range_GHz="1-100", source=ITU),
        dict(material="medium_dry_ground", epsilon=15 * f_ghz ** (-0.1), mu=1, sigma=0.035 * f_ghz ** 1.63,
             roughness=1e-2, frequency_range_GHz="1-10", source=ITU),
    ]


def epsilon_eff(epsilon, sigma, frequency=FREQUENCY):
    """Complex effective relative permittivity of ITU-R P.2040-2 (reference :107-109)."""
    return epsilon - 1j * sigma / (EPSILON_0 * 2 * pi * frequency)


def material_table(frequency=FREQUENCY):
    """Per-material arrays indexed by ``MATERIAL_ID``: (epsilon_eff complex128, roughness float64)."""
    records = material_records(frequency)
    eps = np.array([epsilon_eff(r["epsilon"], r["sigma"], frequency) for r in records], dtype=np.complex128)
    rough = np.array([np.nan if r["roughness"] is None else r["roughness"] for r in records], dtype=np.float64)
    return eps, rough


def _build_dataframe(frequency=FREQUENCY):
    import pandas as pd

    df = pd.DataFrame.from_records(material_records(frequency))
    df["epsilon_eff"] = df["epsilon"] - 1j * df["sigma"] / (EPSILON_0 * 2 * pi * frequency)
    return df


class _LazyFrame:
    """``DF_PROPERTIES`` is built on first use so that importing this module does not import pandas."""

    _df = None

    def _get(self):
        if _LazyFrame._df is None:
            _LazyFrame._df = _build_dataframe()
        return _LazyFrame._df

    def __getattr__(self, name):
        return getattr(self._get(), name)

    def __getitem__(self, item):
        return self._get()[item]


DF_PROPERTIES = _LazyFrame()


def material_of_building_type(building_type):
    """Material name for a building type; unknown types fall back to concrete (reference :123-142)."""
    return _BUILDING_TYPE_TO_MATERIAL.get(building_type, "concrete")


_PROPERTIES_CACHE = {}


def set_properties(building_type, frequency=FREQUENCY):
    """Dict of STRINGS describing the material of a building type, as the reference returns it (:112-143).  Every call
    returns a dict of its own (the strings of a (type, frequency) pair are formed once: a scene asks once per polygon)."""
    material = material_of_building_type(building_type)
    if building_type not in _BUILDING_TYPE_TO_MATERIAL:
        print(f"Properties for the type {building_type} have not been specified, using concrete")
    key = (material, float(frequency))
    cached = _PROPERTIES_CACHE.get(key)
    if cached is None:
        rec = next(r for r in material_records(frequency) if r["material"] == material)
        eps_eff = np.complex128(epsilon_eff(rec["epsilon"], rec["sigma"], frequency))
        cached = _PROPERTIES_CACHE[key] = {
            "material": material,
            "mu": str(np.int64(rec["mu"])),
            "epsilon": str(np.float64(rec["epsilon"])),
            "sigma": str(np.float64(rec["sigma"])),
            "roughness": str(np.float64(rec["roughness"])),
            "epsilon_eff": str(eps_eff),
        }
    return dict(cached)
