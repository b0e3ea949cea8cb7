import sys, os
sys.path.insert(0, os.getcwd())
import alphagpu_amd.lib as aglib
aglib.LIB_PATH = os.path.join(os.getcwd(), 'scratch', sys.argv[1])
sys.argv = [sys.argv[0]] + sys.argv[2:]
exec(open('scratch/prof_search.py').read())
