cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --version | head -1
for fl in "" "-fno-slp-vectorize"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math $fl -o /tmp/slp_reduced slp_reduced.hip 2>/dev/null && echo "flags '$fl': $(/tmp/slp_reduced)"
done
