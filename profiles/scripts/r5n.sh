#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pyramid_only" 2>&1 | tail -40
