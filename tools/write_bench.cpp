// how fast can this box take a file: write() / pwrite() of 8 MiB pieces from n threads (mode 0), mmap + memcpy (mode 2).  tools/e2e_filter_v2_dev.sh
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>
#include <chrono>
static double now(){return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();}
int main(int argc,char**argv){
  const char*path=argv[1]; size_t gb=atoi(argv[2]); int nt=atoi(argv[3]); int mode=atoi(argv[4]);
  size_t n=gb<<30; char*src=(char*)malloc(64<<20); memset(src,'A',64<<20);
  unlink(path); int fd=open(path,O_CREAT|O_TRUNC|O_RDWR,0644);
  double t0=now();
  if(mode==2){ if(ftruncate(fd,n)){perror("ft");return 1;} }
  std::vector<std::thread> th;
  const size_t piece=8<<20; size_t np=n/piece;
  for(int t=0;t<nt;t++) th.emplace_back([&,t]{
    for(size_t p=t;p<np;p+=nt){
      size_t off=p*piece;
      if(mode<2){ size_t done=0; while(done<piece){ ssize_t r=pwrite(fd,src+done,piece-done,off+done); if(r<=0){perror("pw");exit(1);} done+=r; } }
      else { char*m=(char*)mmap(0,piece,PROT_WRITE,MAP_SHARED,fd,off); if(m==MAP_FAILED){perror("mm");exit(1);} memcpy(m,src,piece); munmap(m,piece);} }
  });
  for(auto&x:th)x.join();
  double t1=now(); close(fd);
  printf("%s mode %d threads %d: %.2f GB/s\n",path,mode,nt,n/(t1-t0)/1e9); unlink(path); return 0; }
