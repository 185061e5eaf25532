"""Builds the reference's own native domain-transform sources as CPU torch extensions, from where
they lie under /root/reference, into oracle/_ref/ (git-ignored).  TEST INFRASTRUCTURE ONLY: the
built modules are used by tests/golden/make_golden_native.py to generate golden vectors for the
oracle's restatements (normalized_convolution: NC.cpp:143-204; recursive_filter: RF.cpp:43-92).
Nothing in the product path imports this.  Needs only what the image has: g++ and the torch headers.
"""
import os
import sys

REF = "/root/reference/polyblur/domain_transform"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")


def build(name: str):
    from torch.utils.cpp_extension import load
    src = os.path.join(REF, name + ".cpp")
    if not os.path.exists(src):
        raise FileNotFoundError(src)
    bdir = os.path.join(OUT, name.lower())
    os.makedirs(bdir, exist_ok=True)
    return load(name="polyblur_ref_" + name.lower(), sources=[src], build_directory=bdir,
                extra_cflags=["-O2"], verbose=False)


if __name__ == "__main__":
    for n in (sys.argv[1:] or ["NC", "RF"]):
        m = build(n)
        print(n, "->", m.__file__)
