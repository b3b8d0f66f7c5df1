#!/bin/bash
cd /root/repo
python -X faulthandler -m pytest tests -m gpu -x -q -k "robust" 2>&1 | grep -v "Extension modules" | tail -15 | cut -c1-220
