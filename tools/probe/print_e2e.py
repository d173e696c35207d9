import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); e=d["end_to_end"]; print(sys.argv[1], "events %.2f wall %.2f | host %.2fM dev %.2fM large %.2fM sharder %.2fM"%(d["roofline"]["launch_ms"]*1e3, d["ms_per_step"]*1e3, e["value"]/1e6, e["device_entropy"]["value"]/1e6, e["device_entropy"]["large_batches"]["value"]/1e6, e["sharder"]["value"]/1e6))
