# gpurun with the commit of the tree it sends recorded in GIT_HEAD (the snapshot has no .git): bash tools/gpurun_stamped.sh <timeout> '<command>'
cd "$(dirname "$0")/.."
h=$(git rev-parse --short=12 HEAD)
[ -n "$(git status --porcelain --untracked-files=no)" ] && h="$h-dirty"
echo "$h" > GIT_HEAD
/usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
