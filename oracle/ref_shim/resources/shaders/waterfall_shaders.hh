// TEST INFRASTRUCTURE (oracle/ref_shim): stand-in for the header the reference's build GENERATES from its compiled shader
// binaries (resources/shaders/parser.py).  The compiled-reference checker (oracle/ref_jetstream_build.sh) drives only the
// modules' compute halves (computeSubmit); their present halves -- the only readers of this table -- are never created
// (no Render::Window exists in the harness), so the package is empty.
#pragma once

#include <string>
#include <unordered_map>
#include <vector>

#include "jetstream/memory/types.hh"

static std::unordered_map<std::string, std::unordered_map<Jetstream::DeviceType, std::vector<std::vector<Jetstream::U8>>>> ShadersPackage;
static std::unordered_map<std::string, std::unordered_map<Jetstream::DeviceType, std::vector<std::vector<Jetstream::U8>>>> KernelsPackage;
