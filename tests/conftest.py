import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def pytest_sessionfinish(session, exitstatus):
  # debugging aid: CGS_TEST_DUMP_LIBS=1 lists the librccl / libamdhip64 copies mapped into the process
  if os.environ.get('CGS_TEST_DUMP_LIBS'):
    with open('/proc/self/maps') as f:
      libs = sorted({line.split()[-1] for line in f if 'rccl' in line or 'amdhip' in line or 'libcgsvmc' in line})
    print('\nmapped:', *libs, sep='\n  ')
