#!/bin/sh
# builds tools/host_stage_bench (per-stage cost of the CLI's host side on one thread; needs the built library for the symbols query_cli.cpp refers to)
cd "$(dirname "$0")" && g++ -O3 -std=c++17 -I../include host_stage_bench.cpp ../fmsi_amd/csrc/fastx.cpp ../fmsi_amd/csrc/trace.cpp -o host_stage_bench \
    -L../fmsi_amd/lib -lfmsi_hip -lz -lpthread -ldl -Wl,-rpath,"$(pwd)/../fmsi_amd/lib" && ./host_stage_bench "${1:-20000}" "${2:-/dev/null}"
