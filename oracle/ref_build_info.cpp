// Build-info symbols the reference's common/ expects from its generated build-info.cpp (common/build-info.cpp.in).
// Test infrastructure: part of oracle/_ref/libllama_ref_*.so only.
int LLAMA_BUILD_NUMBER = 0;
char const * LLAMA_COMMIT = "prima.cpp-reference";
char const * LLAMA_COMPILER = "g++";
char const * LLAMA_BUILD_TARGET = "x86_64-linux-gnu (oracle/_ref)";
