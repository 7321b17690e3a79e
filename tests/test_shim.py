"""CPU test of the Haskell shim (integration/haskell/Bulletproofs/Gpu/*.hs).  GHC is not available in the build image,
so the shim cannot be compiled; what CAN be checked is checked here:
  * every `foreign import ccall` names a symbol that include/bpgpu.h declares, with the same number of arguments,
    the Haskell type that corresponds to each C parameter type, and the matching result type;
  * the drop-in functions carry the reference's own type signatures (written out below with their file:line in
    /root/reference, since the reference tree does not travel with the tests)."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "integration", "haskell", "Bulletproofs", "Gpu")

C2HS = {
    "bp_ctx*": "Ptr BpCtx", "const bp_ctx*": "Ptr BpCtx", "bp_ctx**": "Ptr (Ptr BpCtx)",
    "const uint8_t*": "Ptr Word8", "uint8_t*": "Ptr Word8", "uint32_t*": "Ptr Word32",
    "uint32_t": "Word32", "uint64_t": "Word64", "size_t": "CSize", "int": "CInt", "const char*": "CString",
}
RET2HS = {"int": "IO CInt", "const char*": "IO CString", "size_t": "IO CSize", "uint32_t": "IO Word32"}


def c_prototypes():
    text = open(os.path.join(ROOT, "include", "bpgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(bp_[a-z0-9_]+)\s*\(([^;{]*)\)\s*;", text):
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3)
        ret = re.sub(r"\s*\*", "*", ret)
        plist = []
        if params.strip() and params.strip() != "void":
            for p in params.split(","):
                p = " ".join(p.split())
                p = re.sub(r"\s*\*\s*", "* ", p).strip()
                ty = p.rsplit(" ", 1)[0] if " " in p else p   # drop the parameter name
                ty = ty.replace("* *", "**").strip()
                plist.append(re.sub(r"\s+\*", "*", ty))
        protos[name] = (ret, plist)
    return protos


def hs_sources():
    files = sorted(glob.glob(os.path.join(SHIM, "*.hs")))
    assert len(files) >= 6
    return {os.path.basename(f): open(f).read() for f in files}


def split_arrows(sig):
    """Split a Haskell type at top-level `->`."""
    parts, depth, cur, i = [], 0, "", 0
    while i < len(sig):
        c = sig[i]
        if c in "([":
            depth += 1
        elif c in ")]":
            depth -= 1
        if depth == 0 and sig.startswith("->", i):
            parts.append(cur.strip())
            cur = ""
            i += 2
            continue
        cur += c
        i += 1
    parts.append(cur.strip())
    return parts


def foreign_imports():
    out = []
    for fname, text in hs_sources().items():
        text = re.sub(r"--.*$", "", text, flags=re.M)
        for m in re.finditer(r'foreign import ccall\s+(?:safe|unsafe)?\s*"([^"]+)"\s+(\w+)\s*::\s*((?:.|\n)*?)(?=\nforeign import|\n\n|\n[a-zA-Z{-])', text):
            out.append((fname, m.group(1), m.group(2), " ".join(m.group(3).split())))
    return out


def test_every_foreign_import_matches_the_header():
    protos = c_prototypes()
    imports = foreign_imports()
    assert len(imports) >= 19
    for fname, sym, ident, sig in imports:
        assert sym in protos, "%s: %s is not declared in include/bpgpu.h" % (fname, sym)
        ret, params = protos[sym]
        parts = split_arrows(sig)
        args, res = parts[:-1], parts[-1]
        assert len(args) == len(params), "%s: %s has %d arguments, the header has %d" % (fname, sym, len(args), len(params))
        for i, (a, p) in enumerate(zip(args, params)):
            assert p in C2HS, "test table lacks the C type %r (parameter %d of %s)" % (p, i, sym)
            assert a == C2HS[p], "%s: argument %d of %s is %s, the header's %s needs %s" % (fname, i, sym, a, p, C2HS[p])
        assert res == RET2HS[ret], "%s: %s returns %s, the header's %s needs %s" % (fname, sym, res, ret, RET2HS[ret])


def test_the_bindings_cover_the_entry_points_the_drop_in_needs():
    bound = {sym for _, sym, _, _ in foreign_imports()}
    need = {"bp_ctx_create", "bp_generators_derive", "bp_msm", "bp_add_two_mul", "bp_point_mul",
            "bp_rangeproof_prove", "bp_rangeproof_verify_batch", "bp_rangeproof_verify", "bp_ipp_prove", "bp_ipp_verify", "bp_ac_prove",
            "bp_ac_verify", "bp_upper_bound_bits", "bp_log_base2"}
    assert need <= bound, "missing bindings: %s" % sorted(need - bound)


# The reference's signatures (constraints, then argument and result types in order), with where they are defined.
REFERENCE_TYPES = {
    # Bulletproofs/Utils.hs:72, :76, :84
    ("Utils.hs", "sumExps"): "[Fr] -> [PA] -> PA",
    ("Utils.hs", "addTwoMulP"): "Fr -> PA -> Fr -> PA -> PA",
    ("Utils.hs", "commit"): "Fr -> Fr -> PA",
    # Bulletproofs/InnerProductProof/Prover.hs:17-26, Verifier.hs:18-25
    ("InnerProductProof.hs", "generateProof"):
        "InnerProductBase PA -> PA -> InnerProductWitness Fr -> InnerProductProof Fr PA",
    ("InnerProductProof.hs", "verifyProof"):
        "Integer -> InnerProductBase PA -> PA -> InnerProductProof Fr PA -> Bool",
    # Bulletproofs/MultiRangeProof/Prover.hs:19-25, :47-54, Verifier.hs:19-25
    ("MultiRangeProof.hs", "generateProof"):
        "MonadRandom m => Integer -> [(Fr, Fr)] -> ExceptT (RangeProofError Fr) m (RangeProof Fr PA)",
    ("MultiRangeProof.hs", "generateProofUnsafe"):
        "forall m . MonadRandom m => Integer -> [(Fr, Fr)] -> m (RangeProof Fr PA)",
    ("MultiRangeProof.hs", "verifyProof"): "Integer -> [PA] -> RangeProof Fr PA -> Bool",
    # Bulletproofs/RangeProof/Prover.hs:15-20, :25-30, Verifier.hs:17-22
    ("RangeProof.hs", "generateProof"):
        "(MonadRandom m) => Integer -> (Fr, Fr) -> ExceptT (RangeProofError Fr) m (RangeProof Fr PA)",
    ("RangeProof.hs", "generateProofUnsafe"): "(MonadRandom m) => Integer -> (Fr, Fr) -> m (RangeProof Fr PA)",
    ("RangeProof.hs", "verifyProof"): "Integer -> PA -> RangeProof Fr PA -> Bool",
    # Bulletproofs/ArithmeticCircuit/Prover.hs:16-20, Verifier.hs:16-20
    ("ArithmeticCircuit.hs", "generateProof"):
        "forall m . (MonadRandom m) => ArithCircuit Fr -> ArithWitness Fr PA -> m (ArithCircuitProof Fr PA)",
    ("ArithmeticCircuit.hs", "verifyProof"): "[PA] -> ArithCircuitProof Fr PA -> ArithCircuit Fr -> Bool",
}


def test_drop_in_functions_have_the_references_types():
    src = hs_sources()
    for (fname, fn), want in REFERENCE_TYPES.items():
        text = re.sub(r"--.*$", "", src[fname], flags=re.M)
        m = re.search(r"^%s\s*\n?\s*::((?:.|\n)*?)\n%s\b" % (fn, fn), text, flags=re.M)
        assert m, "%s: no top-level signature for %s" % (fname, fn)
        got = " ".join(m.group(1).split())
        assert got == " ".join(want.split()), "%s.%s :: %s, the reference has %s" % (fname, fn, got, want)


def test_range_check_precedes_the_draws():
    """MultiRangeProof/Prover.hs:25-38: the three guards run before anything is drawn from MonadRandom."""
    text = re.sub(r"--.*$", "", hs_sources()["MultiRangeProof.hs"], flags=re.M)
    body = text[text.index("generateProof upperBound vsAndvBlindings = do"):text.index("generateProofUnsafe\n  ::")]
    assert "getRandomR" not in body and "chooseBlindingVectors" not in body
    assert body.index("UpperBoundTooLarge") < body.index("NNotPowerOf2") < body.index("ValuesNotInRange") < \
        body.index("generateProofUnsafe upperBound")
    unsafe = text[text.index("generateProofUnsafe upperBound vsAndvBlindings = do"):]
    order = [unsafe.index(s) for s in ("chooseBlindingVectors nm", "aBlinding  <-", "sBlinding  <-", "t1Blinding <-",
                                       "t2Blinding <-")]
    assert order == sorted(order)


def _code(fname):
    return re.sub(r"--.*$", "", hs_sources()[fname], flags=re.M)


def test_every_context_call_holds_the_process_wide_lock():
    """include/bpgpu.h: a context is not thread-safe.  The drop-ins are pure functions a -threaded program may force
    concurrently, so every foreign call that takes the context runs inside `withCtx` (an MVar), together with the
    error-message read that may follow it; nothing outside FFI.hs can name the raw context."""
    ffi = _code("FFI.hs")
    exports = ffi[ffi.index("module Bulletproofs.Gpu.FFI"):ffi.index(") where")]
    assert "withCtx" in exports and not re.search(r"\bctx\b", exports) and "ctxVar" not in exports
    assert re.search(r"withCtx\s*=\s*withMVar ctxVar", ffi)
    ctx_taking = {ident for _, sym, ident, sig in foreign_imports() if sig.startswith("Ptr BpCtx")}
    assert "c_rangeproof_verify" in ctx_taking and "c_msm" in ctx_taking
    for fname in hs_sources():
        if fname == "FFI.hs":
            continue
        text = _code(fname)
        assert not re.search(r"\bctx\b", text), "%s names the raw context" % fname
        for ident in ctx_taking:
            for m in re.finditer(r"\b%s\b" % ident, text):
                before = text[max(0, m.start() - 40):m.start()]
                assert re.search(r"withCtx \(\\cx -> $", before), "%s: %s is called outside withCtx" % (fname, ident)
                assert text[m.end():m.end() + 4] == " cx ", "%s: %s does not get the locked context" % (fname, ident)


def test_batched_verify_never_sizes_its_buffers_from_a_proof():
    """bp_rangeproof_verify_batch reads 4 + 2 log2(n m) + m rows whatever the proof holds, so the shim
    takes k from the shape (as generateProofUnsafe does), sends only proofs whose lCommits AND rCommits have exactly
    that length and whose commitments pad to the same m, and gives every other proof to the total verifyProof."""
    text = _code("MultiRangeProof.hs")
    body = text[text.index("verifyProofs upperBound batch@"):]
    assert re.search(r"k\s*=\s*log2Ceil \(fromIntegral n \* m\)", body)
    assert "length (lCommits (ipp p)) == k && length (rCommits (ipp p)) == k" in body
    assert "2 ^ log2Ceil (length vs) == m" in body
    assert "isLogBase2 n" in body and "n <= 128" in body
    assert "verifyProof upperBound vs p" in body            # the out-of-shape proofs
    single = text[text.index("verifyProof upperBound vCommits RangeProof{..}"):text.index("verifyProofs\n" if False else "verifyProofs ::")]
    assert "c_rangeproof_verify cx" in single and "min (length lCommits) (length rCommits)" in single
    assert "panic" not in single.replace('panic "verifyProof: negative upper bound', "")  # total: no shape panics
