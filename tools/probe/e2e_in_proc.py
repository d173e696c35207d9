"""probe: does the process around it change the ring's end-to-end rate?  usage: e2e_in_proc.py [torch] [alloc] [hostfirst]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
opts = set(sys.argv[1:])
if "torch" in opts:
    import torch
    torch.cuda.init(); x = torch.zeros(1, device="cuda"); torch.cuda.synchronize()
    if "alloc" in opts:
        keep = [torch.empty(50 << 20, dtype=torch.uint8, device="cuda") for _ in range(8)]
        s = torch.cuda.current_stream()
from lewton_amd import audio, e2e, header, streamgen as sg
setup = sg.stereo_setup(44100, 8, 11)
idp, _, stp = setup.headers()
ident = header.read_header_ident(idp); st = header.read_header_setup(stp, 2, (8, 11))
dec = audio.decoder_for(ident, st, 0)
pool = sg.make_stream(setup, "L", 512, seed=9)
if "hostfirst" in opts:
    r = e2e.measure(dec, pool, 100, 4096, 256, 16, 3, 1)
    print("host stage first: %.2f M" % (r["value"] / 1e6))
for P, nb in ((4096, 300), (16384, 150)):
    r = e2e.measure(dec, pool, nb, P, 256, 0, 3, 1, device_entropy=True)
    print(sorted(opts), "P=%d: %.2f M packets/s, D2H %.1f GB/s" % (P, r["value"] / 1e6, r["d2h_GBps"]))
