#!/bin/bash
# builds and runs tools/micro/load_width.hip (the weight-gradient kernel's access pattern as a pure stream)
cd "$(dirname "$0")/micro" && hipcc -O3 --offload-arch=gfx950 load_width.hip -o load_width 2>/dev/null && ./load_width
