#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 300 python scratch/tail.py 2>&1 | tail -3
