"""TEST INFRASTRUCTURE ONLY: locations of the parity checkers (see oracle/Makefile)."""
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_LIB_PATH = os.path.join(_HERE, "liboracle.so")   # CPU restatement (rtx_oracle.cpp)
REF_LIB_PATH = os.path.join(_HERE, "_ref", "libref.so")  # the reference's own sources + harness
REF_BIN_PATH = os.path.join(_HERE, "_ref", "rt")         # the reference binary, unmodified
REF_RUN_DIR = os.path.join(_HERE, "_ref", "run")
REF_RESOURCES = os.path.join(_HERE, "_ref", "resources")
