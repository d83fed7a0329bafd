#!/usr/bin/env python
"""H36M lifting entry point on MI355X (counterpart of the reference's hpe/main_h36m_lifting.py).
    python hpe/main_h36m_lifting.py train.batch_size=16 train.epochs=2 model.precision=bf16"""
import sys

from _entry import run

if __name__ == "__main__":
    run(sys.argv[1:])
