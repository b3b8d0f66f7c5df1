#!/bin/bash
# one short, bounded GPU command (always under `timeout`)
cd /root/repo
timeout 300 python -X faulthandler -m pytest tests -m gpu -q -k "per_state_policies or stoch or policy or policies or restrict or agents or batched" 2>&1 | grep -v "Extension modules" | tail -40 | cut -c1-260
