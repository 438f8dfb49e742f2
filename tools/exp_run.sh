#!/bin/bash
set -e
Q="timeout -k 10 120 python tools/quick_time.py"
$Q cornell 1024 1024 512
$Q cornell 1024 1024 512 shards 8
$Q cornell 1024 1024 512 schedule 2
