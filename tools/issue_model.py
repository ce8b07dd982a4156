"""SIMD issue-time model of a conv launch from its PMC passes (tools/pmc_conv.sh -> gpurun_out/pmcc_<tag>.txt) and its in-kernel
stamps (STAMPS=1 tools/regb_probe.py): a wave's matrix instruction holds its SIMD for 32 cycles (v_mfma_f32_32x32x16 bf16: 8 passes),
any other vector instruction for 4 (wave64 on 16 lanes); with the two resident workgroups of a CU in phase the two waves of a SIMD add
up.  Prints, per case, vector instructions per MFMA and the modelled workgroup lifetime next to the measured median.
    python tools/issue_model.py <pmcc file> <workgroups> <waves per workgroup> <measured WG cycles> [...]"""
import re
import sys


def load(path):
    d = {}
    for line in open(path):
        m = re.match(r"\s+(\S+)\s+mean\s+([\d.]+)", line)
        if m:
            d[m.group(1)] = float(m.group(2))
    return d


def main():
    a = sys.argv[1:]
    print(f"{'case':28s} {'MFMA/wave':>10s} {'VALU/MFMA':>10s} {'SALU/MFMA':>10s} {'LDS/MFMA':>9s} {'model WG cyc':>13s} {'measured':>9s} {'MFMA share':>11s}")
    for i in range(0, len(a), 4):
        path, wgs, waves, meas = a[i], int(a[i + 1]), int(a[i + 2]), float(a[i + 3])
        c = load(path)
        mf = c["SQ_INSTS_MFMA"]
        va = c["SQ_INSTS_VALU"] - mf                    # (SQ_INSTS_VALU counts the matrix instructions too)
        per_wave = mf / (wgs * waves)
        model = 2 * (32 * per_wave + 4 * va / (wgs * waves))   # two waves per SIMD (two workgroups per CU)
        print(f"{path.split('/')[-1]:28s} {per_wave:10.0f} {va / mf:10.2f} {c['SQ_INSTS_SALU'] / mf:10.2f} {c['SQ_INSTS_LDS'] / mf:9.2f} "
              f"{model:13.0f} {meas:9.0f} {2 * 32 * per_wave / model:11.2f}")


if __name__ == "__main__":
    main()
