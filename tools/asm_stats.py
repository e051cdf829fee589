"""Instruction statistics of one kernel in a hipcc -save-temps .s file (gfx950): totals by class, VGPRs, scratch, and
(optionally) the listing of the kernel.  usage: asm_stats.py file.s substring-of-mangled-name [--dump out.s]"""
import re
import sys
from collections import Counter


def kernels(text):
    for m in re.finditer(r'^(_Z\S+):[^\n]*\n(.*?)^\s*\.end_amdhsa_kernel', text, re.S | re.M):
        yield m.group(1), m.group(2)


def main():
    text = open(sys.argv[1]).read()
    want = sys.argv[2]
    dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
    for name, body in kernels(text):
        if want not in name:
            continue
        code = body.split(".section")[0]
        lines = [l.strip() for l in code.split("\n")]
        ins = [l for l in lines if l and not l.startswith((";", ".")) and not l.endswith(":")]
        c = Counter(i.split()[0] for i in ins)
        def tot(pred):
            return sum(v for k, v in c.items() if pred(k))
        vg = re.search(r"\.amdhsa_next_free_vgpr (\d+)", body)
        sc = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body)
        print(name[:110])
        print("  instructions %d  VALU %d (f64 %d, pk %d, cndmask %d, cvt %d)  SALU %d  vmem loads %d stores %d  lds %d  "
              "waitcnt %d  vgpr %s scratch %s" % (
                  len(ins), tot(lambda k: k.startswith("v_")), tot(lambda k: k.startswith("v_") and "f64" in k),
                  tot(lambda k: k.startswith("v_pk_")), tot(lambda k: k.startswith("v_cndmask")),
                  tot(lambda k: k.startswith("v_cvt")), tot(lambda k: k.startswith("s_") and k != "s_waitcnt"),
                  tot(lambda k: ("buffer_load" in k or "global_load" in k)),
                  tot(lambda k: ("buffer_store" in k or "global_store" in k)), tot(lambda k: k.startswith("ds_")),
                  c["s_waitcnt"], vg.group(1) if vg else "?", sc.group(1) if sc else "?"))
        if dump:
            open(dump, "w").write(code)


if __name__ == "__main__":
    main()
