#!/bin/bash
# Builds a diagnostic copy of the library with in-kernel cycle stamps, runs the probe, restores.
set -e
cd "$GRAFT_REPO_ROOT"
make -C compeg_amd/csrc -s clean
make -C compeg_amd/csrc -s CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-signed-zeros -fvisibility=hidden -DCG_STAMPS"
python tools/stamps_probe.py
make -C compeg_amd/csrc -s clean
make -C compeg_amd/csrc -s
