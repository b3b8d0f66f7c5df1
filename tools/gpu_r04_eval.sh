#!/bin/bash
cd /root/repo
python -m pytest tests -m gpu -x -q -k "batched or evaluation or benchmark" 2>&1 | tail -30
