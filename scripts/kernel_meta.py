"""Register / spill / LDS figures of the gfx950 kernels inside libbm_gar.so (no GPU needed).

    python scripts/kernel_meta.py [regex on the demangled kernel name]

Extracts the code objects with llvm-objdump --offloading into a temporary directory and reads the
AMDGPU metadata notes (llvm-readelf --notes)."""
import pathlib
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = pathlib.Path("/opt/rocm/lib/llvm/bin")
LIB = pathlib.Path(__file__).resolve().parent.parent / "byzantinemomentum_amd" / "libbm_gar.so"


def kernels(lib=LIB):
  out = []
  with tempfile.TemporaryDirectory() as tmp:
    local = pathlib.Path(tmp) / lib.name
    shutil.copy(lib, local)
    subprocess.run([LLVM / "llvm-objdump", "--offloading", local], cwd=tmp, capture_output=True, check=True)
    for co in sorted(pathlib.Path(tmp).glob("*gfx950*")):
      notes = subprocess.run([LLVM / "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
      for block in notes.split("  - .agpr_count:")[1:]:
        def field(name, cast=int):
          m = re.search(r"\." + name + r":\s+(\S+)", block)
          return cast(m.group(1)) if m else None
        out.append({"name": field("name", str), "vgpr": field("vgpr_count"), "sgpr": field("sgpr_count"),
                    "sgpr_spill": field("sgpr_spill_count"), "vgpr_spill": field("vgpr_spill_count"),
                    "lds": field("group_segment_fixed_size"), "scratch": field("private_segment_fixed_size")})
  names = subprocess.run(["c++filt"], input="\n".join(k["name"] for k in out), capture_output=True, text=True).stdout.split("\n")
  for k, nm in zip(out, names):
    k["demangled"] = nm
  return out


def packed_fp32(lib=LIB):
  """{kernel symbol: number of v_pk_{add,mul,fma}_f32 instructions} over the gfx950 code objects of the library — must
  be empty (byzantinemomentum_amd/build.py: the packed forms gave wrong results under GPU sharing)."""
  found = {}
  with tempfile.TemporaryDirectory() as tmp:
    local = pathlib.Path(tmp) / lib.name
    shutil.copy(lib, local)
    subprocess.run([LLVM / "llvm-objdump", "--offloading", local], cwd=tmp, capture_output=True, check=True)
    for co in sorted(pathlib.Path(tmp).glob("*gfx950*")):
      text = subprocess.run([LLVM / "llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
      name = None
      for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
          name = m.group(1)
        elif re.search(r"\bv_pk_(add|mul|fma)_f32\b", line):
          found[name] = found.get(name, 0) + 1
  return found


if __name__ == "__main__":
  pat = re.compile(sys.argv[1]) if len(sys.argv) > 1 else None
  print(f"{'vgpr':>5} {'sgpr':>5} {'s_spill':>7} {'v_spill':>7} {'lds':>7} {'scratch':>7}  kernel")
  for k in kernels():
    if pat is None or pat.search(k["demangled"]):
      print(f"{k['vgpr']:>5} {k['sgpr']:>5} {k['sgpr_spill']:>7} {k['vgpr_spill']:>7} {k['lds']:>7} {k['scratch']:>7}  {k['demangled'][:150]}")
