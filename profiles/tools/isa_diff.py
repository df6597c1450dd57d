"""Compare the instruction streams of the kernels two device assembly listings have in common (labels and comments dropped):
    python profiles/tools/isa_diff.py /tmp/base.s build/obj/trm_launch_column_f64_euler_rich.s
Used to show that moving kernel instantiations between translation units leaves their code unchanged."""
import re
import sys


def kernels(path):
    out, name, body = {}, None, []
    for line in open(path):
        m = re.match(r'^(_Z\S+):', line)
        if m:
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        s = line.strip()
        if s.startswith('.Lfunc_end'):
            out[name] = body
            name = None
            continue
        if line.startswith('\t') and not s.startswith(('.', ';')):
            body.append(re.sub(r'\.LBB\d+_', '.LBB_', re.sub(r'\s*;.*$', '', s)))   # (block labels carry the function's index in its file)
    return out


a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
same = [k for k in b if k in a and a[k] == b[k]]
diff = [k for k in b if k in a and a[k] != b[k]]
print(f"{len(a)} / {len(b)} kernels; common {len(same) + len(diff)}: identical {len(same)}, different {len(diff)}")
for k in diff:
    print("  DIFF", k[:110], len(a[k]), len(b[k]))
