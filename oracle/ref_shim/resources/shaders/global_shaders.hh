// TEST INFRASTRUCTURE (oracle/ref_shim): see spectrogram_shaders.hh -- the generated table of the shaders shared by several
// modules, empty here (no present half is ever created in the compiled-reference checker).
#pragma once

#include <string>
#include <unordered_map>
#include <vector>

#include "jetstream/memory/types.hh"

static std::unordered_map<std::string, std::unordered_map<Jetstream::DeviceType, std::vector<std::vector<Jetstream::U8>>>> GlobalShadersPackage;
static std::unordered_map<std::string, std::unordered_map<Jetstream::DeviceType, std::vector<std::vector<Jetstream::U8>>>> GlobalKernelsPackage;
