# A/B of the 64-row fetch in every non-ML instantiation (-DLM_ROWS64_ALL=1) against the product library: the resident entries and configs[4]
for so in "" tests/tools/ab/lib_r64all.so; do
  echo "== ${so:-product}"
  LORO_AMD_LIB=${so:+$PWD/$so} LM_BINDING_LENIENT=1 python tests/tools/gpu_resident.py 10000 --quick 2>&1 | grep -E "docs_per_s|ms_per_run|integrate" | head -6
  LORO_AMD_LIB=${so:+$PWD/$so} LM_BINDING_LENIENT=1 python tests/tools/gpu_cfg5_shared.py 64 1 2>&1 | tail -1 | cut -c1-420
done
