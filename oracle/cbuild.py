"""Builds oracle/c/ref_kernels.c with gcc into oracle/_build/ (test infrastructure)."""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'c', 'ref_kernels.c')
OUT = os.path.join(HERE, '_build', 'libpvsg_oracle.so')


def build(force=False):
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if force or not os.path.exists(OUT) or os.path.getmtime(OUT) < os.path.getmtime(SRC):
        subprocess.run(['gcc', '-O2', '-shared', '-fPIC', SRC, '-o', OUT, '-lm'], check=True)
    return OUT


def load():
    return ctypes.CDLL(build())
