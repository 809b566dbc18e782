"""Keys of a timing summary.  Same member names, order and ``str()`` as the reference's
``Statistic`` (statistics.py:19-35) so summaries are interchangeable."""
import enum


class Statistic(enum.Enum):
    MIN = enum.auto()
    MAX = enum.auto()
    MED = enum.auto()
    AVG = enum.auto()
    STD = enum.auto()
    NUM = enum.auto()

    # Members are singletons compared by identity, so the identity hash is consistent with ``==`` -- and it is a C
    # slot: ``Enum.__hash__`` is a Python-level ``hash(self._name_)``, paid once per key of every summary dict a
    # report builds (64 sections x 6 statistics per report).
    __hash__ = object.__hash__

    def __str__(self) -> str:
        return self.name

    def __repr__(self) -> str:
        return f"{type(self).__name__}.{self.name}"


#: column of each statistic in the device stats rows (include/nvrx_straggler.h NVRX_STAT_*)
STAT_COLUMNS = (
    (Statistic.MIN, 0),
    (Statistic.MAX, 1),
    (Statistic.MED, 2),
    (Statistic.AVG, 3),
    (Statistic.STD, 4),
    (Statistic.NUM, 5),
)

#: the six keys in column order, and the column of NUM (an integer in the reference's summaries, straggler.py:194)
STAT_KEYS = tuple(stat for stat, _ in STAT_COLUMNS)
NUM_COLUMN = 5
