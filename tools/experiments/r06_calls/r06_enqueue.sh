#!/bin/bash
export TMPDIR=/tmp
python3 tools/cpu_enqueue_time.py full 32 2>/dev/null | tail -1
python3 tools/cpu_enqueue_time.py ssd512 16 2>/dev/null | tail -1
python3 tools/cpu_enqueue_time.py reducedfc 64 fp16 2>/dev/null | tail -1
python3 tools/cpu_enqueue_time.py full 1 2>/dev/null | tail -1
python3 tools/cpu_enqueue_time.py full 4 2>/dev/null | tail -1
