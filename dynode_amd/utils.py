"""The two host helpers of ``dynode.utils`` that the simulate/infer path touches.

Everything else in the reference's utils package (logging, plotting, epiweeks, splines) is outside
the hot path and not rebuilt.
"""

from __future__ import annotations

import datetime
from typing import Any, List


def vectorize_objects(objs: List[Any], target: str, filter: bool = True) -> list:
    """``[getattr(o, target) for o in objs]``, skipping objects without the attribute when
    ``filter`` (reference src/dynode/utils/utils.py:10-38)."""
    if filter:
        return [getattr(o, target) for o in objs if hasattr(o, target)]
    return [getattr(o, target) for o in objs]


def date_to_sim_day(date: datetime.date, init_date: datetime.date) -> int:
    """Days since the model's initialisation date (reference utils/datetime_utils.py)."""
    return (date - init_date).days


def sim_day_to_date(sim_day: int, init_date: datetime.date) -> datetime.date:
    return init_date + datetime.timedelta(days=int(sim_day))
