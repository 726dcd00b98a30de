# Build libgnan_hip_ref.so from the csrc/ of a git revision (default HEAD) next to the working-tree library,
# for same-box A/B runs: bash tools/ab_lib.sh <pkg>/libgnan_hip_ref.so <pkg>/libgnan_hip.so
set -e
REV=${1:-HEAD}
PKG=graph-neural-additive-networks---gnan_amd
T=$(mktemp -d)
mkdir -p $T/csrc $T/include
for f in $(git ls-tree --name-only $REV $PKG/csrc/); do git show $REV:$f > $T/csrc/$(basename $f); done
git show $REV:include/gnan_hip.h > $T/include/gnan_hip.h
for f in $T/csrc/*.hip; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$T/include -I$T/csrc -c $f -o ${f%.hip}.o 2>/dev/null &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC $T/csrc/*.o -o $PKG/libgnan_hip_ref.so
rm -rf $T
echo built $PKG/libgnan_hip_ref.so from $REV
