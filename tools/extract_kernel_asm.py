"""Cut one kernel's body out of a hipcc -S listing:  python tools/extract_kernel_asm.py listing.s 'ILi9ELi1ELi4ELb0ELb0ELb1ELi2ELi1ELb0E' > out.s"""
import sys

path, pat = sys.argv[1], sys.argv[2]
on = False
with open(path) as f:
    for line in f:
        if not on and line.startswith("_ZN") and pat in line and (":" in line):
            on = True
        if on:
            sys.stdout.write(line)
            if line.strip().startswith(".end_amdhsa_kernel") or line.strip() == "s_endpgm" and False:
                break
            if line.startswith(".Lfunc_end"):
                break
