"""Register / spill report of the cooperative policy kernels: compile csrc/policy_wide_kernels.hip with
-Rpass-analysis=kernel-resource-usage (stderr -> argv[1]) and print one line per GRAD / FVP instantiation."""
import re
import sys

txt = open(sys.argv[1]).read()
for b in re.split(r'remark: Function Name: ', txt)[1:]:
    name = b.split()[0]
    m = re.search(r'wide_pass_kernelILi(\d)ELi(\d)ELb(\d)ELi(\d)ELi(\d)E', name)
    if not m:
        continue
    g = lambda k: re.search(re.escape(k) + r': (\d+)', b).group(1)
    L, mode, ks, mt, md = m.groups()
    if mode in ('1', '2'):
        print("L%s %s ksplit%s MT%s MD%s  vgpr %s agpr %s vgpr-spill %s scratch %s B/lane occupancy %s" % (
            L, {'1': 'GRAD', '2': 'FVP '}[mode], ks, mt, md, g('VGPRs'), g('AGPRs'), g('VGPRs Spill'),
            g('ScratchSize [bytes/lane]'), g('Occupancy [waves/SIMD]')))
