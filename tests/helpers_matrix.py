"""Test helper: the full sector matrix H = diag + I (x) H_up + H_dw (x) I as scipy CSR, from the oracle's row lists."""
import numpy as np
import scipy.sparse as sp


def oracle_full_matrix(orc):
    up, dw, d = orc.csr("up"), orc.csr("dw"), orc.diag()
    Hu = sp.csr_matrix((up[2], up[1] - 1, up[0]), shape=(orc.DimUp, orc.DimUp))
    Hd = sp.csr_matrix((dw[2], dw[1] - 1, dw[0]), shape=(orc.DimDw, orc.DimDw))
    Hu.sum_duplicates()
    Hd.sum_duplicates()
    return (sp.diags(d) + sp.kron(sp.identity(orc.DimDw), Hu) + sp.kron(Hd, sp.identity(orc.DimUp))).tocsr()
