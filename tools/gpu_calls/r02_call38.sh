#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_gpu_extensions.py -x -q -m gpu 2>&1 | tail -30
