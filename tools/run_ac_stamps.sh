#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
make -C compeg_amd/csrc -s clean
make -C compeg_amd/csrc -s CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-signed-zeros -fvisibility=hidden -DCG_AC_STAMPS"
python tools/ac_stamps_probe.py
COMPEG_PIPELINE=split python tools/ac_stamps_probe.py
COMPEG_PIPELINE=split COMPEG_WPB=4 COMPEG_LDS_PAD=100000 python tools/ac_stamps_probe.py
COMPEG_PIPELINE=split COMPEG_WPB=1 COMPEG_LDS_PAD=130000 python tools/ac_stamps_probe.py
make -C compeg_amd/csrc -s clean
make -C compeg_amd/csrc -s
