#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export ACE_SFNO_LIB=$PWD/exp/libexp_dbg.so ACE_LIB=$PWD/exp/libexp_dbg.so DBG_LAYERS=2
ACE_NO_CONV_SPLIT=1 python tools/dbg_ws.py run ref 384 180 360 2>&1 | tail -1
ACE_NO_CONV_SPLIT=skip,fc1 python tools/dbg_ws.py run fc2 384 180 360 2>&1 | tail -1
python tools/dbg_ws.py cmp ref fc2 384 180 360 2>&1 | grep -v Warning | grep -A 30 "^P2"
