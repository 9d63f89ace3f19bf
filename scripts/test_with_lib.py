"""A/B aid: run pytest against another build of the library.  usage: test_with_lib.py <lib.so> [pytest args]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cice4_amd import lib
lib.LIBPATH = os.path.abspath(sys.argv[1])
import pytest
raise SystemExit(pytest.main(sys.argv[2:]))
