import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
b=d["berry_loop"]
for key in ("strong","tracking_strong"):
    s=b[key]
    print("=====",key, "fast", s["cholesky_fast_path_fraction"], "seq", round(s["geometries_per_s"]), "lockstep", round(s["lockstep"]["geometries_per_s"]), [round(x,2) for x in s["lockstep"]["step_ms_all_reps"]], "agree", s["lockstep"]["max_abs_energy_difference_vs_sequential"])
    print({k:round(v) for k,v in s["lockstep"].items() if k.endswith("_us")}, {k:round(v) for k,v in s["newton_direction_us"].items() if k not in ("note",)})
    for G,v in s["strong_projection"].items():
        if G!="note": print(G, {k:round(x,3) for k,x in v.items()})
