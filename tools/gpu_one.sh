#!/bin/bash
# one short, bounded GPU command (always under `timeout`)
cd /root/repo
timeout 120 python -X faulthandler -m pytest tests -m gpu -x -q -k "stoch" 2>&1 | grep -v "Extension modules" | tail -6 | cut -c1-220
