"""The C ABI is declared in two headers: include/bt709hip.h (the calls that have a twin in the reference) and
include/bt709hip_ext.h (everything else; it includes the first).  Tests that parse "the header" parse both."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CORE = os.path.join(ROOT, "include", "bt709hip.h")
EXT = os.path.join(ROOT, "include", "bt709hip_ext.h")


def core_text():
    return open(CORE).read()


def ext_text():
    return open(EXT).read()


def text():
    return core_text() + "\n" + ext_text()
