import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cdmft-lanc-ed_amd"))
import hxv
from hxv import models
m = models.hm_2dsquare(Nbath=3)
sec = hxv.HxvSector.from_model(m, 8, 8)
for k in ("slots_in_up_x100","slots_in_dw_x100","slots_out_up_x100","bh_up_x100","rs_up_x100","bh_dw_x100","rs_dw_x100"):
    print(k, sec.get_option(k))
sec.close()
