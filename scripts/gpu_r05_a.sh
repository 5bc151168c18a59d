#!/bin/bash
# round 5, call a: Lua-runtime probe + baseline bench / kbench on today's box
mkdir -p gpurun_out/r05a
{
  echo "== which"; which luajit lua th lua5.1 lua5.3 lua5.4 luarocks qlua 2>&1
  echo "== find"; find / -xdev \( -name 'libluajit*' -o -name 'liblua5*' -o -name 'liblua.*' -o -name 'lua.h' -o -name 'luajit*' -o -name 'lauxlib.h' \) 2>/dev/null | grep -v '^/proc' | head -50
  echo "== python lupa"; python -c 'import lupa; print(lupa.__version__)' 2>&1 | tail -1
  echo "== done"
} > gpurun_out/r05a/lua_probe.txt 2>&1
python bench.py --steps 50 --warmup 10 > gpurun_out/r05a/bench.json 2> gpurun_out/r05a/bench.err
python scripts/kbench.py 128 > gpurun_out/r05a/kbench128.txt 2>&1
cat gpurun_out/r05a/lua_probe.txt; tail -1 gpurun_out/r05a/bench.json | cut -c1-600
