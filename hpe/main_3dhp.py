#!/usr/bin/env python
"""MPI-INF-3DHP lifting entry point on MI355X (counterpart of the reference's hpe/main_3dhp.py with the
conf/data/mpi_inf_3dhp.yaml defaults: 3DHP windows; BASELINE config #5 uses data.seq_len=81)."""
import sys

from _entry import run

if __name__ == "__main__":
    run(sys.argv[1:], extra_defaults={"data.dataset": "3dhp", "data.seq_len": 27, "data.keypoints": "gt"})
