"""Calendar dates as integer simulation days (reference src/dynode/config/simulation_date.py).

The model's initialisation date is published per process in the environment variable
``DYNODE_INITIALIZATION_DATE(<pid>)``; ``simulation_day(y, m, d)`` then gives the (possibly
negative) number of days from it, so configs and priors can be written with calendar dates.
"""

from __future__ import annotations

import datetime
import os
from typing import Optional


def _flag_name() -> str:
    return f"DYNODE_INITIALIZATION_DATE({os.getpid()})"


def set_dynode_init_date_flag(init_date: datetime.date) -> None:
    os.environ[_flag_name()] = init_date.strftime("%Y-%m-%d")


def get_dynode_init_date_flag() -> Optional[datetime.date]:
    """The initialisation date set for this process, or None."""
    text = os.environ.get(_flag_name())
    return None if text is None else datetime.datetime.strptime(text, "%Y-%m-%d").date()


def simulation_day(year: int, month: int, day: int) -> int:
    """Days from the initialisation date to ``date(year, month, day)``; ``ValueError`` if
    `set_dynode_init_date_flag` has not been called in this process."""
    start = get_dynode_init_date_flag()
    if start is None:
        raise ValueError("simulation_day() needs the model's initialisation date: call "
                         "set_dynode_init_date_flag() first")
    return (datetime.date(year, month, day) - start).days
