#!/bin/bash
# Builds a diagnostic copy of the library with wall-clock stamps in the team kernel, runs the probe.
set -e
cd "$GRAFT_REPO_ROOT"
tools/build_variant.sh stamps "-DCG_STAMPS"
COMPEG_LIB=$PWD/gpurun_ab/lib_stamps.so python tools/team_stamps_probe.py
