#!/bin/bash
cd /root/repo
python -X faulthandler -m pytest tests -m gpu -x -q -k "sparse or dense or vi" 2>&1 | grep -v "Extension modules" | tail -6 | cut -c1-220
