#!/bin/bash
cd /root/repo
python -m pytest tests -m gpu -x -q -k "stoch or round3 or subtree or prior" 2>&1 | tail -30
