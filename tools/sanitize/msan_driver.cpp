// MSan driver: a closing grasp + lift, env-steps through the host build of the kernel source (fp32 and fp64)
#include "../../tests/native/ks_lanecheck.cpp"
#include <fstream>
#include <iterator>
int main(int argc, char** argv) {
    FILE* fp = fopen(argv[1], "rb");
    fseek(fp, 0, SEEK_END); long n = ftell(fp); fseek(fp, 0, SEEK_SET);
    unsigned char* blob = (unsigned char*)malloc(n);
    if (fread(blob, 1, n, fp) != (size_t)n) return 2;
    fclose(fp);
    void* h = lc_create(blob, n);
    if (!h) return 1;
    for (int prec : {32, 64}) {
        for (int start = 0; start < 3; start++) {
            double qpos[16] = {0}, qvel[15] = {0}, warm[15] = {0}, hq[4] = {0.5, -0.5, -0.5, -0.5}, obs[82], rew, rays[17];
            int done;
            qpos[9] = 0.03 * (start - 1); qpos[10] = 0.01; qpos[11] = 0.0654; qpos[12] = 1;
            if (argc > 4) { qpos[9] += atof(argv[2]); qpos[10] += atof(argv[3]); qpos[11] = atof(argv[4]); }   // object offset (multi-geom pieces carry their CAD origin)
            if (start == 2) { hq[0] = 1; hq[1] = hq[2] = hq[3] = 0; qpos[2] = -0.05; }     // hand flat near the ground
            lc_reset_obs(h, prec, qpos, qvel, warm, hq, obs, &rew, &done, rays);
            for (int t = 0; t < 24; t++) {
                double act[4] = {t > 14 ? 0.6 : 0.0, 0.5, 0.6, 0.7};
                int st = lc_env_step(h, prec, qpos, qvel, warm, hq, act, 15, 6, obs, &rew, &done, rays);
                if (t % 8 == 7) std::printf("prec %d start %d t %d status %d z %.5f obs0 %.4f\n", prec, start, t, st, qpos[11], obs[0]);
            }
        }
    }
    lc_destroy(h);
    return 0;
}
