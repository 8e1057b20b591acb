#!/usr/bin/env python3
"""Round-5 study infrastructure (NOT product, NOT test): a scratch copy of the oracle under /tmp/dbg/oracle with debug hooks, used by
tools/r05/row22_*.py, row46_param.py, replay_from.py.  Hooks (globals set through ctypes): ko_trace (MPR branch / box-support tie trace on
stderr), ko_dbg_drop_g1/g2 (skip a pair), ko_dbg_qfrc[15] (applied generalized force), ko_dbg_g1 (hand geom whose contact with the object
the next hooks act on), ko_dbg_ddist / rotz / tilt / dpos[3] / mu / Rscale / frot (perturb that contact), ko_dbg_obj_first (0 = the
operand order of rounds 1-4), ko_dbg_skew, ko_dbg_pnormal (portal normal instead of libccd's witness direction), ko_dbg_mjterm / mjtol
(MuJoCo's Newton stop rule).  usage: python tools/r05/make_debug_oracle.py"""
import shutil, subprocess
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
DST = Path("/tmp/dbg/oracle")
if DST.exists():
    shutil.rmtree(DST)
shutil.copytree(ROOT / "oracle", DST, ignore=shutil.ignore_patterns("*.so", "__pycache__", "_ref"))
p = DST / "ko_physics.c"
s = p.read_text()

def rep(old, new, count=1):
    global s
    assert old in s, old[:60]
    s = s.replace(old, new, count)

rep('#include <string.h>', '''#include <string.h>
#include <stdio.h>
int ko_trace = 0, ko_dbg_drop_g1 = -1, ko_dbg_drop_g2 = -1, ko_dbg_g1 = 3, ko_dbg_obj_first = 1, ko_dbg_pnormal = 0, ko_dbg_mjterm = 0;
double ko_dbg_qfrc[15], ko_dbg_ddist = 0, ko_dbg_rotz = 0, ko_dbg_dpos[3] = {0, 0, 0}, ko_dbg_tilt = 0, ko_dbg_mu = 0, ko_dbg_Rscale = 1, ko_dbg_frot = 0,
       ko_dbg_skew = 1e-6, ko_dbg_mjtol = 1e-8;
int ko_dbg_tie_mode = 0, ko_dbg_tie_n = 0, ko_dbg_tie_script[256] = {0}; /* mode 1: every near-zero component of a box support direction takes its sign from the next script entry */
int ko_dbg_partner = 0, ko_dbg_tie_code[16] = {0}; /* tie study: sign pattern (bits 0-2 flip x y z) of the box's support skew, per partner geom */''')
rep('const double sk = KO_SUPPORT_SKEW * (fabs(ld[0])', 'const double sk = ko_dbg_skew * (fabs(ld[0])')
rep('ld[0] += sk * KO_SKEW_X; ld[1] += sk * KO_SKEW_Y; ld[2] += sk * KO_SKEW_Z;',
    '''if (ko_dbg_tie_mode == 1 && g == 8) {
          for (int k = 0; k < 3; k++) if (fabs(ld[k]) < 1e-9 * (fabs(ld[0]) + fabs(ld[1]) + fabs(ld[2]))) { ld[k] = (ko_dbg_tie_script[ko_dbg_tie_n & 255] ? -1e-12 : 1e-12); ko_dbg_tie_n++; }
      } else
      { const int code = (g == 8) ? ko_dbg_tie_code[ko_dbg_partner & 15] : 0;
          ld[0] += sk * KO_SKEW_X * ((code & 1) ? -1 : 1); ld[1] += sk * KO_SKEW_Y * ((code & 2) ? -1 : 1); ld[2] += sk * KO_SKEW_Z * ((code & 4) ? -1 : 1); }''')
rep('    mulmatTvec3(ld, s->geom_xmat[g], dir);', '''    mulmatTvec3(ld, s->geom_xmat[g], dir);
    if (ko_trace >= 2 && g == 8) { double mn = fabs(ld[0]); if (fabs(ld[1]) < mn) mn = fabs(ld[1]); if (fabs(ld[2]) < mn) mn = fabs(ld[2]);
        fprintf(stderr, "      box support local dir (%.3e %.3e %.3e)%s\\n", ld[0], ld[1], ld[2], mn < 1e-9 ? "  <-- TIE" : ""); }''')
rep('''            if (vec_is_zero(wit)) return -1; /* normal undefined */
            copy3(dir, wit);
            normalize3(dir);''', '''            if (ko_trace) { double pn[3], wn[3] = {wit[0], wit[1], wit[2]}; portal_dir(&v1, &v2, &v3, pn); normalize3(wn);
                fprintf(stderr, "   MPR g%d-g%d pen_it %d depth %.6e portal_n (%.6f %.6f %.6f) wit_n (%.6f %.6f %.6f) cos %.12f depth_along_n %.6e\\n", c->g1, c->g2, it,
                        *depth, pn[0], pn[1], pn[2], wn[0], wn[1], wn[2], dot3(pn, wn), dot3(pn, v1.v)); }
            if (vec_is_zero(wit)) return -1; /* normal undefined */
            copy3(dir, wit);
            normalize3(dir);
            if (ko_dbg_pnormal) { double pn[3]; portal_dir(&v1, &v2, &v3, pn); copy3(dir, pn); if (ko_dbg_pnormal == 1) *depth = dot3(pn, v1.v); }''')
rep('''    copy3(c->pos, pos);
    copy3(c->frame, normal);
    make_frame(c->frame);''', '''    copy3(c->pos, pos);
    copy3(c->frame, normal);
    if (g1 == ko_dbg_g1 && g2 == 8) {
        c->dist += ko_dbg_ddist;
        for (int i = 0; i < 3; i++) c->pos[i] += ko_dbg_dpos[i];
        double cz = cos(ko_dbg_rotz), sz = sin(ko_dbg_rotz), nx = c->frame[0], ny = c->frame[1];
        c->frame[0] = cz * nx - sz * ny; c->frame[1] = sz * nx + cz * ny;
        c->frame[2] += ko_dbg_tilt;
        if (ko_dbg_mu > 0) c->mu[0] = c->mu[1] = ko_dbg_mu;
    }
    make_frame(c->frame);
    if (ko_dbg_frot != 0 && g1 == ko_dbg_g1 && g2 == 8) {
        double cz = cos(ko_dbg_frot), sz = sin(ko_dbg_frot), t1[3], t2[3];
        for (int i = 0; i < 3; i++) { t1[i] = cz * c->frame[3 + i] + sz * c->frame[6 + i]; t2[i] = -sz * c->frame[3 + i] + cz * c->frame[6 + i]; }
        for (int i = 0; i < 3; i++) { c->frame[3 + i] = t1[i]; c->frame[6 + i] = t2[i]; }
    }''')
rep('const int obj_first = (g2 == KO_OBJ_GEOM);', 'const int obj_first = ko_dbg_obj_first && (g2 == KO_OBJ_GEOM);')
rep('    const int obj_first = ko_dbg_obj_first && (g2 == KO_OBJ_GEOM);', '    ko_dbg_partner = g1;\n    const int obj_first = ko_dbg_obj_first && (g2 == KO_OBJ_GEOM);')
rep('''        int g1 = (int)m->pairs[p][0], g2 = (int)m->pairs[p][1];
        if (g1 == 0)''', '''        int g1 = (int)m->pairs[p][0], g2 = (int)m->pairs[p][1];
        if (g1 == ko_dbg_drop_g1 && g2 == ko_dbg_drop_g2) continue;
        if (g1 == 0)''')
rep('add_row(s, 2, J, c->dist, c->margin, diag);', 'add_row(s, 2, J, c->dist, c->margin, (c->geom1 == ko_dbg_g1 && c->geom2 == 8) ? diag * ko_dbg_Rscale : diag);')
rep('''f[i] = s->qfrc_passive[i] - s->qfrc_bias[i] + s->qfrc_actuator[i];
    chol_solve(s->L, f, s->qacc_smooth);''', '''f[i] = s->qfrc_passive[i] - s->qfrc_bias[i] + s->qfrc_actuator[i] + ko_dbg_qfrc[i];
    chol_solve(s->L, f, s->qacc_smooth);''')
rep('f[i] = s->qfrc_passive[i] - s->qfrc_bias[i] + s->qfrc_actuator[i] + s->qfrc_constraint[i];',
    'f[i] = s->qfrc_passive[i] - s->qfrc_bias[i] + s->qfrc_actuator[i] + s->qfrc_constraint[i] + ko_dbg_qfrc[i];')
rep('''        s->newton_iters_used = it + 1;
        s->newton_last_step = dmax / (1 + amax);''', '''        s->newton_iters_used = it + 1;
        s->newton_last_step = dmax / (1 + amax);
        if (ko_dbg_mjterm) { /* MuJoCo: stop when scale * (oldcost - cost) or scale * |grad| < tolerance, scale = 1 / (meaninertia * nv) */
            double mi = 0; int nd = 0;
            for (int i = 0; i < KO_NV; i++) if (s->M[i][i] < 1e6) { mi += s->M[i][i]; nd++; }
            double scale = 1.0 / (mi / nd * nd), anew[KO_NV], aold[KO_NV], gg[KO_NV], gn2 = 0;
            for (int i = 0; i < KO_NV; i++) { anew[i] = a[i]; aold[i] = a[i] - alpha * p[i]; }
            double c0 = primal_cost(s, aold), c1 = primal_cost(s, anew);
            for (int i = 0; i < KO_NV; i++) { gg[i] = -qfrc_smooth[i]; for (int j = 0; j < KO_NV; j++) gg[i] += s->M[i][j] * anew[j]; }
            for (int i = 0; i < n; i++) { double x = -s->efc_aref[i]; for (int k = 0; k < KO_NV; k++) x += s->efc_J[i][k] * anew[k];
                if (s->efc_type[i] == 0 || x < 0) for (int k = 0; k < KO_NV; k++) gg[k] += s->efc_J[i][k] * x / s->efc_R[i]; }
            for (int k = 0; k < KO_NV; k++) gn2 += gg[k] * gg[k];
            if (scale * (c0 - c1) < ko_dbg_mjtol || scale * sqrt(gn2) < ko_dbg_mjtol) { s->newton_converged = 2; break; }
            continue;
        }''')
p.write_text(s)
subprocess.check_call(["make", "-C", str(DST), "-s"])
print("built", DST / "libko_oracle.so")
