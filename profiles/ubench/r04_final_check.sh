# last check of the round: build from a clean tree state, smoke(), the whole GPU suite, the default bench line's four legs
timeout -k 10 400 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -E "smoke OK|AssertionError|Error" | cut -c1-200
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/gputests.log 2>&1; echo rc=$?; tail -2 gpurun_out/gputests.log
