"""Run a command as a child process and report its peak resident set: python tools/maxrss.py <cmd...> (exit code = the child's)."""
import resource
import subprocess
import sys

rc = subprocess.call(sys.argv[1:])
print("maxrss of children: %.1f GB" % (resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss / 1e6), flush=True)
sys.exit(rc)
