#!/bin/bash
cd /root/repo
python -m pytest tests -m gpu -x -q -k "opd and not saopd" 2>&1 | tail -8
