"""Small array helpers of the path-solve path.

Mirrors the two helpers of the reference's ``raytracing/array_utils.py`` that the hot path uses
(``normalize`` :5, ``sort_by_columns`` :9-41).
"""
import numpy as np


def normalize(array, axis=None):
    """Divide by the Euclidean norm (reference ``array_utils.py:5``)."""
    array = np.asarray(array, dtype=float)
    return array / np.linalg.norm(array, axis=axis)


def sort_by_columns(array, columns=None):
    """Sort the rows of a 2-D array lexicographically, first column most significant.

    This is what defines ``edge[0]`` of a sharp edge: the endpoint that is smaller in
    (x, then y, then z) order (reference ``array_utils.py:9-41``, used at ``geometry.py:551-557``).
    """
    array = np.asarray(array)
    if columns is None:
        columns = range(array.shape[1])
    keys = [array[:, c] for c in reversed(list(columns))]
    return array[np.lexsort(keys)]
