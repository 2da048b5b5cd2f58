#!/usr/bin/env python3
"""Dev: run a script under a watchdog that dumps every thread's Python stack after N seconds (hang triage).
usage: dbg_run.py SECONDS script.py [args...]"""
import faulthandler
import runpy
import sys

secs = float(sys.argv[1])
script = sys.argv[2]
sys.argv = sys.argv[2:]
faulthandler.dump_traceback_later(secs, exit=True)
runpy.run_path(script, run_name="__main__")
