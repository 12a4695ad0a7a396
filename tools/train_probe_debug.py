#!/usr/bin/env python
"""debug wrapper: tools/train_step_probe.py with a stack dump if it is still running after 45 s"""
import faulthandler
import runpy
import sys

faulthandler.dump_traceback_later(45, exit=True)
sys.argv = [sys.argv[0]] + sys.argv[1:]
runpy.run_path("tools/train_step_probe.py", run_name="__main__")
