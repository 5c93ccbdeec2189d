import sys
sys.path.insert(0, "cdmft-lanc-ed_amd")
import torch, hxv
from hxv import models
for name, m, sec_ in (("C3", models.hm_2dsquare(Nbath=3), (8, 8)), ("C4", models.bhz_2d(Nbath=1), (8, 8))):
    sec = hxv.HxvSector.from_model(m, *sec_)
    g = sec.get_option
    print(name, sec.stats())
    print("  bits", g("tile_bits_up"), g("tile_bits_dw"), "blocks", g("nblocks_up"), g("nblocks_dw"), "maxblock", g("max_block_up"), g("max_block_dw"),
          "slots in up/dw", g("slots_in_up_x100") / 100, g("slots_in_dw_x100") / 100, "bh/rs up", g("bh_up_x100") / 100, g("rs_up_x100") / 100,
          "dw", g("bh_dw_x100") / 100, g("rs_dw_x100") / 100, "n_in/out up", g("n_in_up"), g("n_out_up"), "dw", g("n_in_dw"), g("n_out_dw"))
    v = torch.randn(sec.fullElems, dtype=torch.float64, device="cuda") + 1j * torch.randn(sec.fullElems, dtype=torch.float64, device="cuda")
    hv = torch.empty_like(v)
    for p in (1, 2, 3):
        sec.set_option("passes", p); sec.time_apply(v, hv, 1); print("  passes", p, "ms", round(sec.time_apply(v, hv, 5), 3))
