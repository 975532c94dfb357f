"""tools/ that time or check the lanes-over-the-batch kernel run the DEVELOPMENT build of the library (the product library does
not contain that kernel any more): call use_dev_library() before the first victor_amd call of the process."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def use_dev_library():
    sys.path.insert(0, ROOT)
    from victor_amd.build import build_native
    os.environ["VICTOR_HIP_LIB"] = build_native(dev=True)
    os.environ["VICTOR_HIP_DEV"] = "1"
