from .meters import (AverageValueMeter, EpochResultDict, MeterInterface, MeterResultDict, MultipleAverageValueMeter,  # noqa: F401
                     SurfaceMeter, UniversalDice)
from .storage import Storage, StorageIncomeDict  # noqa: F401
