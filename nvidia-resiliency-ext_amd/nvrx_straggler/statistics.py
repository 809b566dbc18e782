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
