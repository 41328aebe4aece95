#!/bin/bash
export BTS_LIB_PATH=$GRAFT_REPO_ROOT/ab/laptrace.so
python scripts/dbg_jobtrace.py target c5 2>&1 | grep -v amdgpu.ids
python scripts/dbg_mid2trace.py target 2>&1 | grep -v amdgpu.ids
