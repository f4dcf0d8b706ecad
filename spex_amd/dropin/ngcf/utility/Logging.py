"""print + append to a log file (NGCF_SPEX/code/utility/Logging.py)."""
from spex_amd.dropin.utility1.Logging import Logging  # noqa: F401
