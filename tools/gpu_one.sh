#!/bin/bash
# one short, bounded GPU command (always under `timeout`)
cd /root/repo
timeout 400 python -m pytest tests -m gpu -q 2>&1 | tail -3 | cut -c1-250
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-120
