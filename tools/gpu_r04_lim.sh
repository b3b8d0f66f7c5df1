#!/bin/bash
cd /root/repo
python -m pytest tests -m gpu -x -q -k "state_aware or saopd or robust or lds_atomics" 2>&1 | tail -5
