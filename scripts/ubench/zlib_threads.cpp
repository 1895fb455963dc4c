#include <thread>
#include <vector>
#include <chrono>
#include <cstdio>
#include <zlib.h>
#include <cstring>
int main(){
  const int NB=1000; std::vector<std::vector<unsigned char>> comp(NB); std::vector<unsigned char> src(30000);
  for (size_t i=0;i<src.size();i++) src[i]=(unsigned char)((i*2654435761u)>>13);
  for (int b=0;b<NB;b++){ uLongf cl=compressBound(30000); comp[b].resize(cl); compress2(comp[b].data(), &cl, src.data(), 30000, 6); comp[b].resize(cl);}
  for (int nt : {1,2,4,8}) {
    auto t0=std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t=0;t<nt;t++) th.emplace_back([&,t]{ std::vector<unsigned char> out(30000); for (int rep=0;rep<4;rep++) for (int b=NB*t/nt;b<NB*(t+1)/nt;b++){ uLongf ol=30000; uncompress(out.data(), &ol, comp[b].data(), comp[b].size()); } });
    for (auto&x:th) x.join();
    printf("%d threads %.3f s\n", nt, std::chrono::duration<double>(std::chrono::steady_clock::now()-t0).count());
  }
}
