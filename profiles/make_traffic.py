"""fetch/write PMC summaries (summarize_rocpd.py pmc) -> traffic json read by bench.py.

  python profiles/make_traffic.py profiles/r1e_pmc_fetch.csv profiles/r1e_pmc_write.csv 1024000 profiles/r1e_traffic.json [merged_by_atoms.json]

bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE/WRITE_SIZE are in KiB, and on gfx950
FETCH_SIZE tallies the 128-byte fabric read requests at 64 bytes (MI355X_MICROARCH.md, HBM section).
"""
import csv
import json
import sys

NAMES = [("CheckGatherBody", "gather_skin_check"), ("RadialWin2Body", "radial_descriptor"), ("RadialWinBody", "radial_descriptor"),
         ("ForceWinBody", "force_assemble"),
         ("ResidentStepBody", "velocity_verlet"), ("RadialTileBody", "radial_descriptor"),
         ("RadialDescBody", "radial_descriptor"), ("AngularFusedBody", "angular_fused"), ("AngularDescBody", "angular_descriptor"),
         ("nepmi_ann_mfma", "ann"), ("AnnBody", "ann"), ("AngularForceBody", "angular_partial_force"),
         ("ForceTileBody", "force_assemble"), ("ForceAssembleBody", "force_assemble"), ("VerletSeamBody", "velocity_verlet"),
         ("VelocityVerletBody", "velocity_verlet_unfused")]


def read(path, counter):
    """first matching NAMES entry wins per bench name, whatever the row order of the csv (e.g. the
    window force assembly over the record-reading one, which only runs while the engine picks)"""
    rows = []
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["counter"] == counter:
                rows.append(row)
    out = {}
    # the scatter form of the force assembly is two launches (window scatter + fold): their sum is the force assembly
    def first(key):
        # (of the scatter kernel's variants the one without energy / virial outputs comes first: the step between two records)
        cand = sorted((row for row in rows if key in row["kernel"]), key=lambda r: -int(r["dispatches"]))
        return float(cand[0]["sum_per_dispatch"]) if cand else None
    sc, fo = first("nepmi_force_scatter"), first("ForceFoldBody")
    if sc is not None and fo is not None:
        out["force_assemble"] = sc + fo
    for key, name in NAMES:
        if name in out:
            continue
        # of several instantiations of a body (list forms, output variants) the one the steps ran most often
        cand = sorted((row for row in rows if key in row["kernel"] and "nepmi_fused_image" not in row["kernel"]),  # (the one-off image
                      key=lambda r: -int(r["dispatches"]))                                                       #  builder is not the kernel)
        if cand:
            out[name] = float(cand[0]["sum_per_dispatch"])
    return out


def main():
    fetch, write = read(sys.argv[1], "FETCH_SIZE"), read(sys.argv[2], "WRITE_SIZE")
    kern = {}
    for name in fetch:
        w = write.get(name, 0.0)
        kern[name] = {"fetch_kb": fetch[name], "write_kb": w, "hbm_bytes_per_launch": (2.0 * fetch[name] + w) * 1024.0}
    entry = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `bench.py --steps 6 --warmup 2`; "
                       "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch (gfx950 FETCH_SIZE correction, "
                       "MI355X_MICROARCH.md)", "atoms": int(sys.argv[3]), "kernels": kern}
    # --merge FILE (6th argument): add this entry to a by_atoms file (one entry per workload size: bench.py: load_traffic)
    if len(sys.argv) > 5:
        import os
        merged = json.load(open(sys.argv[5])) if os.path.exists(sys.argv[5]) else {}
        if "by_atoms" not in merged:
            merged = {"source": entry["source"], "by_atoms": {}}
        merged["by_atoms"][str(entry["atoms"])] = entry
        json.dump(merged, open(sys.argv[5], "w"), indent=1)
    json.dump(entry, open(sys.argv[4], "w"), indent=1)


if __name__ == "__main__":
    main()
