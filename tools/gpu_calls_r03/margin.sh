#!/bin/bash
python tools/margin_check.py 2>&1 | grep -v amdgpu.ids
