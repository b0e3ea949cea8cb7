#!/usr/bin/env python3
"""CLI of tests/spill_exec_check.py (spill code at the head of a block in front of an exec restore): scratch/spill_exec_check.py [objects...]"""
import os, runpy, sys
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "spill_exec_check.py"), run_name="__main__")
