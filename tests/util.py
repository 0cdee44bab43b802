"""Shared helpers for the parity tests (tests only; may import oracle/)."""
import ctypes as C
import os

import numpy as np

import oracle as O
from xcltk_amd import capi
from xcltk_amd.engine import Engine

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_tables(d):
    regions = O.load_regions(os.path.join(d, "regions.tsv"))
    snp_fn = os.path.join(d, "snps.tsv")
    snps = O.load_snps(snp_fn) if os.path.isfile(snp_fn) else []
    return regions, snps


def batch_from_dict(g):
    return capi.make_batch(g["contig"], g["ordinal_base"], g["pos"], g["flag"], g["mapq"], g["cell"],
                           g["umi"], g["cig_off"], g["cigar"], g.get("seq_off"), g.get("seq"))


def engine_vs_oracle(mode, names, regions, snps, n_cells, batches, flags=0, **filt):
    """Run the HIP engine and the C oracle on the same SoA batches; return both COO dicts."""
    kw = dict(min_mapq=20, min_len=30, incl_flag=0, excl_flag=772, no_orphan=True, min_include=0.9,
              min_count=1, min_maf=0, no_dup_hap=True)
    kw.update(filt)
    eng = Engine(mode, names, regions, n_cells, snps=snps if mode == capi.XCK_MODE_BAF else (),
                 flags=flags, **kw)
    try:
        for b, _ in batches:
            eng.push(b)
        got = eng.finish()
        stats = eng.stats()
    finally:
        eng.close()
    cfg, keep = O.make_config(mode, names, regions, snps if mode == capi.XCK_MODE_BAF else [], n_cells,
                              flags=flags, **kw)
    exp = O.run_oracle(cfg, [b for b, _ in batches])
    return got, exp, stats


def assert_coo_equal(got, exp, names):
    for k in names:
        g, e = got[k], exp[k]
        assert len(g[0]) == len(e[0]), "%s: nnz %d != %d" % (k, len(g[0]), len(e[0]))
        for j, what in enumerate(("row", "col", "val")):
            assert np.array_equal(g[j], e[j]), "%s.%s differs" % (k, what)


# ----------------------------------------------------------------------------- golden cases
import json


def list_cases():
    cdir = os.path.join(GOLDEN, "cases")
    return sorted(os.listdir(cdir)) if os.path.isdir(cdir) else []


def load_case(name, tmpdir):
    """-> (case dict with substituted paths, dataset dir, out dir, expected dir)"""
    cdir = os.path.join(GOLDEN, "cases", name)
    with open(os.path.join(cdir, "case.json")) as fp:
        case = json.load(fp)
    ddir = os.path.join(GOLDEN, "datasets", case["dataset"])
    with open(os.path.join(ddir, "dataset.json")) as fp:
        ds = json.load(fp)
    odir = os.path.join(str(tmpdir), "out_" + name)
    listfile = os.path.join(str(tmpdir), "bam_list_%s.txt" % name)
    with open(listfile, "w") as fp:
        fp.write("".join(os.path.join(ddir, b) + "\n" for b in ds["bams"]))

    def sub(v):
        if isinstance(v, str):
            return v.replace("$D/", ddir + "/").replace("$O", odir).replace("$L", listfile)
        return v
    if "argv" in case:
        case["argv"] = [sub(a) for a in case["argv"]]
    else:
        case["kwargs"] = {k: sub(v) for k, v in case["kwargs"].items()}
    return case, ddir, odir, os.path.join(cdir, "expected")


def oracle_params(case):
    """Translate a case (wrapper kwargs or basefc argv) into oracle.run_files() arguments,
    applying the defaults of the reference wrappers (rdr/fc/main.py:142-182, baf/fc/main.py:32-77)."""
    mode = capi.XCK_MODE_BASEFC if case["kind"] == "basefc" else capi.XCK_MODE_BAF
    if "argv" in case:
        a = case["argv"]
        kw = {}
        names = {"-s": "sam_fn", "-b": "barcode_fn", "-R": "region_fn", "-O": "out_dir",
                 "--exclFLAG": "excl_flag", "--inclFLAG": "incl_flag", "--minMAPQ": "min_mapq",
                 "--minINCLUDE": "min_include", "-p": "ncores", "--minLEN": "min_len"}
        i = 0
        while i < len(a):
            k = names[a[i]]
            v = a[i + 1]
            if k in ("excl_flag", "incl_flag", "ncores", "min_len"):
                v = int(v)
            elif k == "min_mapq":
                v = float(v)
            elif k == "min_include":
                v = float(v) if "." in v else int(v)
            kw[k] = v
            i += 2
        explicit_excl = kw.get("excl_flag")
    else:
        kw = dict(case["kwargs"])
        # fc_wrapper never forwards a non-None excl_flag (reference quirk); afc_wrapper does
        explicit_excl = kw.get("excl_flag") if mode == capi.XCK_MODE_BAF else None
    if kw.get("sam_fn"):
        bams = kw["sam_fn"].split(",")
    else:
        with open(kw["sam_list_fn"]) as fp:
            bams = [x.rstrip() for x in fp]
    sample_ids = None
    if not kw.get("barcode_fn"):
        with open(kw["sample_id_fn"]) as fp:
            sample_ids = [x.strip() for x in fp]
    p = dict(mode=mode, bam_fns=bams, region_fn=kw["region_fn"], barcode_fn=kw.get("barcode_fn"),
             sample_ids=sample_ids, snp_fn=kw.get("phased_snp_fn"), cell_tag=kw.get("cell_tag", "CB"),
             umi_tag=kw.get("umi_tag", "UB"), excl_flag=explicit_excl,
             output_all_reg=kw.get("output_all_reg", mode == capi.XCK_MODE_BASEFC),
             min_mapq=kw.get("min_mapq", 20), min_len=kw.get("min_len", 30), incl_flag=kw.get("incl_flag", 0),
             no_orphan=kw.get("no_orphan", True))
    if mode == capi.XCK_MODE_BASEFC:
        p["min_include"] = kw.get("min_include", 0.9)
    else:
        p.update(min_count=kw.get("min_count", 1), min_maf=kw.get("min_maf", 0), no_dup_hap=kw.get("no_dup_hap", True))
        if kw.get("cellsnp_dir"):
            p["phase"] = phase_from_cellsnp(kw["cellsnp_dir"], kw.get("ref_cell_fn"), p["output_all_reg"])
    return p


def phase_from_cellsnp(cellsnp_dir, ref_cell_fn, output_all_reg):
    """Region-wise local phasing for the oracle runs: the product's HOST code (xcltk_amd/baf/fc/phasing.py - float64 numpy,
    no GPU) decides the final haplotype indices and the dropped (region, SNP) pairs; the oracle then does the counting.
    Against the reference's golden outputs this pins that host code on machines without a GPU."""
    from xcltk_amd.baf.fc.main import regions_with_snps
    from xcltk_amd.baf.fc.phasing import local_phasing
    from xcltk_amd.utils.csp_io import load_data

    def phase(regions, snps):
        csp = load_data(cellsnp_dir)
        ref_cells = None
        if ref_cell_fn:
            with open(ref_cell_fn) as fp:
                ref_cells = [x.strip() for x in fp if x.strip()]
        has = regions_with_snps(regions, snps)
        idx = list(range(len(regions))) if output_all_reg else [i for i, h in enumerate(has) if h]
        rh, ah, er, es, _ = local_phasing([regions[i] for i in idx], snps, csp, ref_cells)
        snps2 = [(s[0], s[1], s[2], s[3], int(r), int(a)) for s, r, a in zip(snps, rh.tolist(), ah.tolist())]
        return snps2, (np.array([idx[i] for i in er.tolist()], dtype=np.int32), es)
    return phase


def assert_dirs_equal(got_dir, exp_dir):
    exp = sorted(os.listdir(exp_dir))
    got = sorted(f for f in os.listdir(got_dir) if not f.startswith("."))
    assert got == exp, (got, exp)
    for f in exp:
        a = open(os.path.join(got_dir, f), "rb").read()
        b = open(os.path.join(exp_dir, f), "rb").read()
        assert a == b, "%s differs from the reference output" % f
