"""Many-keys-to-one-value dictionary used for sharp edges.

Same observable behaviour as the reference's ``raytracing/container_utils.py:24-48``:
keys are tuples of polygon indices; ``d[i, j, ..., Ellipsis]`` returns every ``(key, value)`` whose key
contains any of the listed indices, in insertion order.  The reference scans all keys on every such
lookup; this one keeps an index → keys incidence table so the lookup is proportional to the answer.
"""
import json

import numpy as np


class ManyToOneDict(dict):
    def __init__(self, *args, **kwargs):
        super().__init__()
        self._order = {}      # key -> insertion rank
        self._incidence = {}  # member index -> list of keys containing it
        if args or kwargs:
            for key, value in dict(*args, **kwargs).items():
                self[key] = value

    def _register(self, key):
        if key not in self._order:
            self._order[key] = len(self._order)
            for member in key:
                self._incidence.setdefault(member, []).append(key)

    def __setitem__(self, key, value):
        if not isinstance(key, tuple):
            key = (key,)
        elif len(key) > 0 and key[-1] is Ellipsis:
            for match, _ in self[key]:
                dict.__setitem__(self, match, value)
            # the reference then also stores the value under the ellipsis key itself
        self._register(key)
        dict.__setitem__(self, key, value)

    def _matching(self, members):
        seen = set()
        for member in members:
            for key in self._incidence.get(member, ()):
                seen.add(key)
        return sorted(seen, key=self._order.__getitem__)

    def __getitem__(self, item):
        if not isinstance(item, tuple):
            return [(key, dict.__getitem__(self, key)) for key in self._matching([item])]
        if len(item) > 0 and item[-1] is Ellipsis:
            return [(key, dict.__getitem__(self, key)) for key in self._matching(item[:-1])]
        return item, dict.__getitem__(self, item)

    def __eq__(self, other):
        if not isinstance(other, ManyToOneDict) or len(self) != len(other):
            return False
        for key, value in self.items():
            if key not in other:
                return False
            other_value = dict.__getitem__(other, key)
            if isinstance(value, np.ndarray) and isinstance(other_value, np.ndarray):
                if not np.array_equal(value, other_value):
                    return False
            elif value != other_value:
                return False
        return True

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None

    def to_json(self, filename=None):
        data = {}
        for key, value in self.items():
            data[str(tuple(int(k) for k in key))] = value.tolist() if isinstance(value, np.ndarray) else value
        if filename is not None:
            with open(filename, "w") as f:
                json.dump(data, f)
        return json.dumps(data)

    @classmethod
    def from_json(cls, data=None, filename=None):
        if data is None and filename is not None:
            with open(filename) as f:
                data = json.load(f)
        if data is None:
            raise ValueError("Either data or filename must be provided")
        if isinstance(data, str):
            data = json.loads(data)
        out = cls()
        for key, value in data.items():
            key_tuple = tuple(int(x) for x in filter(None, key.strip("()").replace(" ", "").split(",")))
            out[key_tuple] = np.array(value) if isinstance(value, list) else value
        return out
