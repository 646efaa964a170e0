"""``infos`` of a batched step, built on demand.

The VecEnv protocol hands back one dict per env and step (`infos[i]` = the reference's per-step ``info`` dict, plus SB3's
``terminal_observation`` / ``TimeLimit.truncated`` for envs that finished).  Building 4096 dicts costs ~4 ms of Python per step --
thirty times the kernel, the transfers and everything else in ``step()`` together -- while the consumers of ``infos`` (SB3's rollout
collection, ``VecMonitor``, the reference's ``RewardCallback``) look at the finished envs or at a few entries.  ``LazyInfos`` IS a
``list`` (``isinstance`` checks, ``len``, indexing, slicing, iteration, in-place edits of ``infos[i]`` all behave) whose dicts come
into existence when they are first touched; what is never touched is never built.  Operations of ``list`` that work on the raw
storage as a whole (``+`` from either side, ``*``, ``pop``, ``sort``, ``reverse``, ``remove``, ``insert``, ``append`` / ``extend``,
``del``, slice assignment) first build every dict, so none of them can hand out or move around an unbuilt slot.  The arrays of the step (component rows,
terminal observations) are captured, not copied per env.
"""
from __future__ import annotations

import types

# what `infos[i]` is for an env that did not finish when the env was built with infos_mode="finished": one shared, read-only, empty
# mapping (`.get`, `in`, iteration, `.copy()` behave; assignment raises instead of leaking into every other env's info)
NO_INFO = types.MappingProxyType({})


def finished_only_infos(n: int, finished: dict) -> list:
    """Plain list for ``infos_mode="finished"``: full dicts for the envs in ``finished`` (index -> dict), ``NO_INFO`` elsewhere.
    SB3's rollout collection calls ``info.get("episode")`` on EVERY element every step, which would materialise every lazily built
    dict (~1 us each); with this mode that loop runs over one shared empty mapping and only finished envs carry content."""
    infos = [NO_INFO] * n
    for i, d in finished.items():
        infos[i] = d
    return infos


class LazyInfos(list):
    """``list`` of per-env info dicts; ``make(i)`` builds the dict of env ``i`` the first time it is asked for."""

    def __init__(self, n: int, make):
        super().__init__([None] * n)
        self._make = make

    # -- element access -------------------------------------------------------------------------------------------------------
    def _get(self, i: int):
        d = list.__getitem__(self, i)
        if d is None:
            n = list.__len__(self)
            d = self._make(i + n if i < 0 else i)
            list.__setitem__(self, i, d)
        return d

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._get(j) for j in range(*i.indices(list.__len__(self)))]
        return self._get(i)

    def __iter__(self):
        return (self._get(i) for i in range(list.__len__(self)))

    def __reversed__(self):
        return (self._get(i) for i in range(list.__len__(self) - 1, -1, -1))

    def __contains__(self, item):
        return any(d == item for d in self)

    # -- whole-list operations: on the materialised list ----------------------------------------------------------------------------
    def materialize(self) -> list:
        return [self._get(i) for i in range(list.__len__(self))]

    def copy(self):
        return self.materialize()

    def __eq__(self, other):
        return self.materialize() == (other.materialize() if isinstance(other, LazyInfos) else other)

    def __ne__(self, other):
        return not self == other

    __hash__ = None

    def __repr__(self):
        return repr(self.materialize())

    def __add__(self, other):
        return self.materialize() + list(other)

    def __reduce__(self):                       # pickles / deep-copies as a plain list (the builder is a closure)
        return (list, (self.materialize(),))

    def index(self, item, *args):
        return self.materialize().index(item, *args)

    def count(self, item):
        return self.materialize().count(item)

    # -- inherited operations that work on the raw storage: build everything first, then let ``list`` do it --------------------------
    def _fill(self):
        for i in range(list.__len__(self)):
            self._get(i)

    def __radd__(self, other):
        return list(other) + self.materialize()

    def __mul__(self, k):
        return self.materialize() * k

    __rmul__ = __mul__

    def __iadd__(self, other):
        self._fill()
        list.extend(self, other)
        return self

    def __imul__(self, k):
        self._fill()
        return list.__imul__(self, k)

    def pop(self, *args):
        self._fill()
        return list.pop(self, *args)

    def remove(self, item):
        self._fill()
        list.remove(self, item)

    def insert(self, i, item):
        self._fill()
        list.insert(self, i, item)

    def append(self, item):
        self._fill()
        list.append(self, item)

    def extend(self, other):
        self._fill()
        list.extend(self, other)

    def sort(self, *args, **kw):
        self._fill()
        list.sort(self, *args, **kw)

    def reverse(self):
        self._fill()
        list.reverse(self)

    def __delitem__(self, i):
        self._fill()
        list.__delitem__(self, i)

    def __setitem__(self, i, v):
        if isinstance(i, slice):
            self._fill()
        list.__setitem__(self, i, v)

    def clear(self):
        list.clear(self)
