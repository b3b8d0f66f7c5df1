#!/bin/bash
cd /root/repo
python -m pytest tests -m gpu -x -q -k "robust_planner" 2>&1 | tail -8
