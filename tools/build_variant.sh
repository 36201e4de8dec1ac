# Build a library variant for same-box A/B runs (tools/ab_variants.sh, ODBOX_LIBRARY):
#   bash tools/build_variant.sh <name> <hipcc flags...>      -> object_detection_amd/build/libodbox_<name>.so
# The in-tree libodbox.so is rebuilt with the default flags afterwards.
set -e
name=$1; shift
ODBOX_EXTRA_HIPCC_FLAGS="$*" python -m object_detection_amd.build --force > /dev/null
cp object_detection_amd/libodbox.so object_detection_amd/build/libodbox_$name.so
python -m object_detection_amd.build --force > /dev/null
echo object_detection_amd/build/libodbox_$name.so
