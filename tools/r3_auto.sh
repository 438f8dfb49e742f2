#!/bin/bash
# What AUTO picks now, timed.
set -e
T="timeout -k 10"
q() { $T 300 python3 tools/quick_time.py "$@"; }
q final 1920 1080 256
q final 1920 1080 64
q random 1920 1080 64
q final 1920 1080 16
q cornell 1024 1024 4096
