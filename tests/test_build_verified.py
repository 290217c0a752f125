"""The register-allocator miscompile of round 4 (csrc/Makefile: MEM_CLAUSE; NOTES R4.2) is only visible to LLVM's machine verifier, and a
verifier build of the kernels takes a quarter of an hour -- too long for this suite.  So the check is split: `make -C eqf_vio_amd/csrc verify`
compiles the device code of the kernel-carrying translation units with `-mllvm -verify-machineinstrs` (plus the two -mllvm flags of the
product build), fails on "Bad machine code" and records the hash of the sources it saw in csrc/VERIFIED (committed); this test recomputes the
hash.  A source change that has not been through the verifier fails here."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "eqf_vio_amd", "csrc")


def source_hash():
    import glob
    import hashlib

    names = sorted([os.path.basename(p) for p in glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp"))]
                   + ["../../include/eqf_vio_amd.h", "../../include/eqf_vio_amd_debug.h", "Makefile"])  # the Makefile's $(sort ...) order
    h = hashlib.sha256()
    for n in names:
        with open(os.path.join(CSRC, n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def test_kernel_sources_have_been_through_the_machine_verifier():
    path = os.path.join(CSRC, "VERIFIED")
    assert os.path.exists(path), "run `make -C eqf_vio_amd/csrc verify` (about 20 minutes) and commit csrc/VERIFIED"
    seen = open(path).read().strip()
    assert seen == source_hash(), ("the kernel sources changed since the last `-verify-machineinstrs` build: run `make -C eqf_vio_amd/csrc verify` "
                                   "(about 20 minutes, fails on 'Bad machine code') and commit csrc/VERIFIED")


def test_verify_target_uses_the_product_flags():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    assert "-verify-machineinstrs" in mk and "$(CXXFLAGS) $(VERIFY_FLAGS)" in mk  # same -O3 / -mllvm flags as the product objects
    assert "Bad machine code" in mk
