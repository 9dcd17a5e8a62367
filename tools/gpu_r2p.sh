#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 120 tools/repro/trans_then_pk_f32 2>&1 | tail -12
