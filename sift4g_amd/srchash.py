"""Hash of the scan kernels' source: what a stored rocprofv3 counter summary (profiles/hbm_traffic.json, profiles/valu_executed.json)
is keyed on.  bench.py copies `traffic` / `executed` out of a committed profile only while the sources the profiled build was made
from are the sources of the build that is running; tools/summarize_profile.py writes the hash next to the numbers."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCAN_SOURCES = ("sift4g_amd/csrc/sw_kernels.hip.h", "sift4g_amd/csrc/s4g_scan.hip")


def kernel_source_sha(root=ROOT, files=SCAN_SOURCES):
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(root, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read() + b"\0")
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernel_source_sha())
