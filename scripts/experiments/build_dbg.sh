#!/bin/bash
# The diagnostic library (phase stamps: -DBN_TILE_CLOCK) -> build/libbn_dbg.so; sources copied so that the product objects stay untouched.
set -e
cd "$(dirname "$0")/../.."
mkdir -p build/dbg_csrc
cp bayesiannetwork_amd/csrc/*.hip bayesiannetwork_amd/csrc/*.hpp bayesiannetwork_amd/csrc/*.cpp bayesiannetwork_amd/csrc/Makefile build/dbg_csrc/
sed -i 's#\.\./\.\./include/#../../include/#' build/dbg_csrc/*.hpp build/dbg_csrc/*.cpp 2>/dev/null || true
make -C build/dbg_csrc -j8 EXTRA="-DBN_TILE_CLOCK ${DBG_EXTRA:-}" OUT=../libbn_dbg.so > build/dbg_build.log 2>&1 || { tail -20 build/dbg_build.log; exit 1; }
ls -la build/libbn_dbg.so
