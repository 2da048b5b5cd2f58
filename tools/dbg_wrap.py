#!/usr/bin/env python3
"""Debug aid: run a script and dump every thread's Python stack after DUMP_AFTER seconds (then exit) - for ranks that hang.
usage: python tools/dbg_wrap.py script.py args..."""
import faulthandler
import os
import runpy
import sys

faulthandler.dump_traceback_later(int(os.environ.get("DUMP_AFTER", "30")), exit=True)
sys.argv = sys.argv[1:]
sys.path.insert(0, os.path.dirname(os.path.abspath(sys.argv[0])))
runpy.run_path(sys.argv[0], run_name="__main__")
