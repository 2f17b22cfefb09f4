"""The boundary's struct layouts, held together from three sides: include/midas_snps.h (what the library is compiled
against), midas_amd/abi.py (the tested binding) and the ctypes stubs INTEGRATION.md shows a MIDAS maintainer.

A C program generated from the header prints sizeof / offsetof / field size of every public struct; every ctypes.Structure
of the binding and of INTEGRATION.md's ```python blocks (executed against a stand-in CDLL) must agree field by field, in
name, order, offset and size.  A struct added to the header without a binding, or a field added to one side only, fails here.
"""
import ctypes as C
import os
import re
import subprocess
import sys
import types

import pytest

from midas_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "midas_snps.h")

# header struct -> the binding's class
ABI_CLASSES = {
    "midas_snps_thresholds": abi.Thresholds,
    "midas_snps_reads": abi._Reads,
    "midas_snps_contigs": abi._Contigs,
    "midas_snps_batch_info": abi.BatchInfo,
    "midas_merge_params": abi.MergeParams,
    "midas_merge_genes": abi._Genes,
}
# INTEGRATION.md's class names -> header struct
DOC_CLASSES = {"Thresholds": "midas_snps_thresholds", "Reads": "midas_snps_reads", "Contigs": "midas_snps_contigs",
               "MergeParams": "midas_merge_params", "BatchInfo": "midas_snps_batch_info", "Genes": "midas_merge_genes"}


def header_structs():
    """{struct name: [field names in order]} of every `typedef struct X { ... } X;` with a body in the header."""
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        assert m.group(1) == m.group(3)
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if decl:
                fields.append(re.search(r"(\w+)\s*(\[\w*\])?$", decl).group(1))
        out[m.group(1)] = fields
    return out


@pytest.fixture(scope="module")
def c_layouts(tmp_path_factory):
    """{struct: (sizeof, [(field, offset, size)])} as gcc lays the header out."""
    structs = header_structs()
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "midas_snps.h"', 'int main(void) {']
    for s, fields in structs.items():
        lines.append('  printf("S %s %%zu\\n", sizeof(%s));' % (s, s))
        for f in fields:
            lines.append('  printf("F %s %s %%zu %%zu\\n", offsetof(%s, %s), sizeof(((%s*)0)->%s));' % (s, f, s, f, s, f))
    lines += ['  printf("V %d\\n", MIDAS_SNPS_ABI_VERSION);', '  return 0;', '}']
    d = tmp_path_factory.mktemp("layouts")
    src = d / "layouts.c"
    src.write_text("\n".join(lines) + "\n")
    exe = d / "layouts"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                   check=True)
    text = subprocess.run([str(exe)], check=True, stdout=subprocess.PIPE, text=True).stdout
    out, version = {}, None
    for ln in text.splitlines():
        p = ln.split()
        if p[0] == "S":
            out[p[1]] = [int(p[2]), []]
        elif p[0] == "F":
            out[p[1]][1].append((p[2], int(p[3]), int(p[4])))
        else:
            version = int(p[1])
    return {k: (v[0], v[1]) for k, v in out.items()}, version


def ctypes_layout(cls):
    return C.sizeof(cls), [(n, getattr(cls, n).offset, getattr(cls, n).size) for n, _ in cls._fields_]


def test_the_header_is_plain_c_and_every_struct_has_a_binding(c_layouts):
    layouts, version = c_layouts
    assert version == abi.ABI_VERSION
    assert sorted(layouts) == sorted(ABI_CLASSES), "a public struct of include/midas_snps.h has no ctypes class in abi.py (or vice versa)"


@pytest.mark.parametrize("name", sorted(ABI_CLASSES))
def test_binding_struct_matches_the_header(c_layouts, name):
    layouts, _ = c_layouts
    assert ctypes_layout(ABI_CLASSES[name]) == layouts[name]


def integration_blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    return re.findall(r"```python\n(.*?)```", text, flags=re.S)


def exec_blocks():
    """Load INTEGRATION.md's stubs the way a maintainer's modules would load -- in order, `C.CDLL("libmidas_snps_hip.so")`
    resolving to the library built in this tree, section B importable as midas.run.hip_pileup -- and return their
    namespaces."""
    from midas_amd import build
    real = C.CDLL(build.build_native())
    fake = types.ModuleType("ctypes")
    fake.__dict__.update({k: getattr(C, k) for k in dir(C) if not k.startswith("__")})
    fake.CDLL = lambda *a, **k: real
    saved = {k: sys.modules.get(k) for k in ("ctypes", "midas", "midas.run", "midas.run.hip_pileup")}
    sys.modules["ctypes"] = fake
    out = []
    try:
        for k, code in enumerate(integration_blocks()):
            ns = {"__name__": "integration_stub_%d" % k}
            exec(compile(code, "INTEGRATION.md", "exec"), ns)
            out.append(ns)
            if k == 0:
                for name in ("midas", "midas.run", "midas.run.hip_pileup"):
                    sys.modules[name] = types.ModuleType(name)
                sys.modules["midas.run.hip_pileup"].__dict__.update(ns)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return out


def test_integration_stubs_match_the_header(c_layouts):
    layouts, _ = c_layouts
    seen = set()
    for ns in exec_blocks():
        for name, obj in ns.items():
            if isinstance(obj, type) and issubclass(obj, C.Structure) and obj is not C.Structure:
                assert name in DOC_CLASSES, "INTEGRATION.md defines a struct this test does not know: %s" % name
                assert ctypes_layout(obj) == layouts[DOC_CLASSES[name]], \
                    "INTEGRATION.md's %s does not lay out like %s in include/midas_snps.h" % (name, DOC_CLASSES[name])
                seen.add(name)
    assert {"Thresholds", "Reads", "Contigs", "MergeParams"} <= seen


def test_integration_stubs_are_complete_python():
    """Every name a stub uses at module level or inside its functions is imported or defined by some stub above it (the
    maintainer pastes them in order), and the pileup stub refuses a library of another ABI version."""
    import builtins
    import symtable
    known = set(dir(builtins))
    blocks = integration_blocks()
    assert len(blocks) >= 3
    for code in blocks:
        top = symtable.symtable(code, "INTEGRATION.md", "exec")
        defined = {s.get_name() for s in top.get_symbols() if s.is_assigned() or s.is_imported() or s.is_namespace()}
        known |= defined

        def walk(tab):
            for s in tab.get_symbols():
                if s.is_global() and s.is_referenced() and not s.is_assigned():
                    assert s.get_name() in known, "INTEGRATION.md: name %r is used but never imported or defined" % s.get_name()
            for ch in tab.get_children():
                walk(ch)
        for s in top.get_symbols():
            if s.is_referenced() and not (s.is_assigned() or s.is_imported() or s.is_namespace()):
                assert s.get_name() in known, "INTEGRATION.md: name %r is used but never imported or defined" % s.get_name()
        for ch in top.get_children():
            walk(ch)
    assert "midas_snps_abi_version" in blocks[0]


def _stub_table(ns, contigs):
    import numpy as np
    c = ns["Contigs"](contigs.n_contigs, contigs.n_species, contigs.length.ctypes.data, contigs.species.ctypes.data,
                      contigs.read_begin.ctypes.data, contigs.ref.ctypes.data, None)
    return types.SimpleNamespace(c=c, n_sites=int(np.sum(contigs.length)), keep=contigs)


def test_integration_bam_stub_reads_what_the_binding_reads(tmp_path):
    """Section B's load_bam, pasted as it stands, against the library on the host decoder: same columns as abi.read_bam."""
    import numpy as np
    from midas_amd import bam, synth
    contigs, reads = synth.make_dataset(n_species=1, contigs_per_species=2, contig_len=5000, n_reads=400, seed=11, var_len=True)
    names = ["c%d" % i for i in range(contigs.n_contigs)]
    refid = np.repeat(np.arange(contigs.n_contigs, dtype=np.int32), np.diff(contigs.read_begin))
    path = str(tmp_path / "s.bam")
    bam.write_bam(path, names, contigs.length.tolist(), refid, reads)
    ns = exec_blocks()[0]
    h, got_names, got_refid, r = ns["load_bam"](path)
    try:
        assert got_names == names and np.array_equal(got_refid, refid) and r.n_reads == reads.n_reads
        view = lambda ptr, dt, n: np.ctypeslib.as_array(C.cast(ptr, C.POINTER(dt)), (n,))
        assert np.array_equal(view(r.pos, C.c_int32, reads.n_reads), reads.pos)
        assert np.array_equal(view(r.nm, C.c_int32, reads.n_reads), reads.nm)
        assert np.array_equal(view(r.qual_off, C.c_int64, reads.n_reads + 1), reads.qual_off)
        assert np.array_equal(view(r.qual, C.c_uint8, int(reads.qual_off[-1])), reads.qual)
        assert np.array_equal(view(r.cigar, C.c_uint32, int(reads.cigar_off[-1])), reads.cigar)
    finally:
        ns["lib"].midas_bam_close.argtypes = [C.c_void_p]
        ns["lib"].midas_bam_close(h)


@pytest.mark.gpu
def test_integration_pileup_stub_counts_what_the_oracle_counts(tmp_path):
    """Section B end to end on the GPU, as a maintainer would run it: BAM -> load_bam -> hip_count_coverage."""
    import numpy as np
    from midas_amd import bam, synth
    from oracle import c_oracle
    contigs, reads = synth.make_dataset(n_species=2, contigs_per_species=2, contig_len=7001, n_reads=3000, seed=12, var_len=True)
    names = ["c%d" % i for i in range(contigs.n_contigs)]
    refid = np.repeat(np.arange(contigs.n_contigs, dtype=np.int32), np.diff(contigs.read_begin))
    path = str(tmp_path / "s.bam")
    bam.write_bam(path, names, contigs.length.tolist(), refid, reads)
    ns = exec_blocks()[0]
    h, _, _, r = ns["load_bam"](path)
    counts, allele, stats = ns["hip_count_coverage"](abi.DEFAULT_ARGS, _stub_table(ns, contigs), r)
    st, _, oc, oa, os_ = c_oracle.pileup(abi.Thresholds.from_args(abi.DEFAULT_ARGS), contigs, reads)
    assert st == 0 and np.array_equal(counts, oc) and np.array_equal(allele, oa) and np.array_equal(stats, os_)
