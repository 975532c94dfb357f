"""Run a Python script under a watchdog: with_watchdog.py SECONDS script.py [args...] - after SECONDS the tracebacks of all
threads are written to stderr and the process exits (a hung rendezvous or collective then says where it hangs)."""
import faulthandler
import runpy
import sys

seconds = float(sys.argv[1])
faulthandler.dump_traceback_later(seconds, exit=True)
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
